// FRI folding over F_{p^3}.  No reference counterpart in /root/reference; serves
// the GenChunkProof request of src/prover/provider.rs:358-377.
//
// f is given by its evaluations on the coset shift*<w_n> (natural order, three planes u64[3][n]).
// Folding by 2^logf:  f(x) = sum_j x^j g_j(x^(2^logf)),  out(y) = sum_j beta^j g_j(y).  For output
// index i (m = n >> logf) the 2^logf inputs i + m*k are the values of h_i(z) = sum_j g_j(y_i) z^j on
// the coset x_i*<w_(2^logf)>, x_i = shift*w_n^i:  inverse DFT of size 2^logf in registers per plane
// (the twiddles are base-field), undo x_i^j, Horner at beta.  lane = output index, so all three
// plane reads per k are coalesced runs.
#include <hip/hip_runtime.h>

#include "ctx.hpp"
#include "wave_xchg.hpp"

namespace {

struct FoldArgs {
    const u64 *in;
    u64 *out;
    const u64 *twl, *twh;  // w_n^-e two-level table
    u64 w16[8];            // w_16^-i
    u64 beta[3];
    u64 shift_inv, finv;
    int logn, lb;
};

__device__ __forceinline__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

template <int A>
__device__ __forceinline__ void dif(u64 *v, const u64 *w16) {
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int half = 1 << (A - 1 - s);
#pragma unroll
        for (int b = 0; b < (1 << A); b += 2 * half) {
#pragma unroll
            for (int i = 0; i < half; i++) {
                u64 x = v[b + i], y = v[b + i + half];
                v[b + i] = gl_add(x, y);
                u64 d = gl_sub(x, y);
                const int e = i * (8 / half);
                v[b + i + half] = (e == 0) ? d : gl_mul(d, w16[e]);
            }
        }
    }
}

template <int LOGF>
__global__ void __launch_bounds__(256) fri_fold_kernel(FoldArgs a) {
    constexpr int F = 1 << LOGF;
    const u64 n = 1ULL << a.logn, m = n >> LOGF;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    u64 v[3][F];
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int k = 0; k < F; k++) v[c][k] = a.in[(u64)c * n + i + m * k];
        dif<LOGF>(v[c], a.w16);  // v[c][p] = F * cof_{brev(p)}
    }
    // x_i^-1 = shift^-1 * w_n^-i
    const u64 wi = gl_mul(a.twl[i & ((1ULL << a.lb) - 1)], a.twh[i >> a.lb]);
    const u64 xinv = gl_mul(a.shift_inv, wi);
    // gamma = beta * x_i^-1 ;  result = finv * sum_j cof_j gamma^j   (Horner from the top)
    e3 gamma = e3_scale(e3_make(a.beta[0], a.beta[1], a.beta[2]), xinv);
    e3 acc = e3_make(0, 0, 0);
#pragma unroll
    for (int j = F - 1; j >= 0; j--) {
        const int p = brev(j, LOGF);
        acc = e3_mul(acc, gamma);
        acc = e3_add(acc, e3_make(v[0][p], v[1][p], v[2][p]));
    }
    acc = e3_scale(acc, a.finv);
#pragma unroll
    for (int c = 0; c < 3; c++) a.out[(u64)c * m + i] = acc.c[c];
}

// The fold by 16 with a coset SPREAD OVER THE 16 LANES OF A DPP ROW (zp_set_tuning "fri_fold_lanes"; the A/B against the register form above is
// profiles/r5_dpp_ab.txt): lane k of a row holds input i + m k, the four DIF levels are lane exchanges (ds_swizzle 8, 4; DPP quad permutes 2, 1),
// every lane raises gamma to ITS exponent and the sixteen terms are summed by four more exchanges.  Same result, bit for bit; what it costs:
// a row reads sixteen addresses m apart (four outputs per wave: 8-byte accesses where the register form reads 512-byte runs), both sides of
// every butterfly run in every lane, and the powers of gamma are 5-7 extension products per lane against ONE per input in Horner's form.
template <int LH>
__device__ __forceinline__ u64 fold_stage(u64 x, int kk, const u64 *tws) {
    const u64 y = lane_xor<LH>(x);
    const int i = kk & ((1 << LH) - 1);
    if (!((kk >> LH) & 1)) return gl_add(x, y);
    const u64 d = gl_sub(y, x);
    return i ? gl_mul(d, tws[(u64)(i * (8 >> LH)) << 8]) : d;          // w_16^-e = w_4096^-(256 e), e = i 8 / half
}

__global__ void __launch_bounds__(256) fri_fold_lanes_kernel(FoldArgs a, const u64 *__restrict__ tws) {
    const u64 n = 1ULL << a.logn, m = n >> 4;
    const int kk = threadIdx.x & 15;
    const u64 i0 = ((u64)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool on = i0 < m;
    const u64 i = on ? i0 : m - 1;                       // idle rows follow the exchanges on a valid address and store nothing
    u64 v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        u64 x = a.in[(u64)c * n + i + m * (u64)kk];
        x = fold_stage<3>(x, kk, tws);
        x = fold_stage<2>(x, kk, tws);
        x = fold_stage<1>(x, kk, tws);
        x = fold_stage<0>(x, kk, tws);
        v[c] = x;                                        // 16 cof_{brev(kk)}, plane c
    }
    const u64 wi = gl_mul(a.twl[i & ((1ULL << a.lb) - 1)], a.twh[i >> a.lb]);
    const e3 gamma = e3_scale(e3_make(a.beta[0], a.beta[1], a.beta[2]), gl_mul(a.shift_inv, wi));
    const int j = brev(kk, 4);
    e3 pw = e3_make(1, 0, 0), g = gamma;                 // gamma^j by squaring
#pragma unroll
    for (int bit = 0; bit < 4; bit++) {
        if ((j >> bit) & 1) pw = e3_mul(pw, g);
        if (bit < 3) g = e3_mul(g, g);
    }
    e3 term = e3_mul(e3_make(v[0], v[1], v[2]), pw);
#pragma unroll
    for (int c = 0; c < 3; c++) {                        // sum over the row
        u64 x = term.c[c];
        x = gl_add(x, lane_xor<3>(x));
        x = gl_add(x, lane_xor<2>(x));
        x = gl_add(x, lane_xor<1>(x));
        x = gl_add(x, lane_xor<0>(x));
        term.c[c] = x;
    }
    if (on && kk == 0) {
        const e3 r = e3_scale(term, a.finv);
#pragma unroll
        for (int c = 0; c < 3; c++) a.out[(u64)c * m + i0] = r.c[c];
    }
}

}  // namespace

extern "C" int32_t zp_fri_fold(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t logf,
                               const uint64_t beta[3], uint64_t shift) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "fri_fold");
    ZP_ARG(ctx, logf >= 1 && logf <= 4, "logf must be in 1..4");
    ZP_ARG(ctx, logn >= logf && logn <= 32, "logn out of range");
    ZP_ARG(ctx, d_in && d_out && beta, "null pointer");
    ZP_ARG(ctx, d_in != d_out, "fold cannot run in place");
    if (shift == 0) shift = ctx->coset_shift;
    ZP_ARG(ctx, shift < GL_P && beta[0] < GL_P && beta[1] < GL_P && beta[2] < GL_P, "not canonical");
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logn, true, &pl));
    FoldArgs a;
    a.in = (const u64 *)d_in;
    a.out = (u64 *)d_out;
    a.twl = pl->d_twl;
    a.twh = pl->d_twh;
    a.lb = pl->lb;
    a.logn = logn;
    for (int i = 0; i < 8; i++) a.w16[i] = pl->w16[i];
    for (int i = 0; i < 3; i++) a.beta[i] = beta[i];
    a.shift_inv = gl_inv(shift);
    a.finv = gl_inv(1ULL << logf);
    const u64 m = 1ULL << (logn - logf);
    dim3 grid((unsigned)((m + 255) / 256)), block(256);
    // DPP rows where they measured faster (profiles/r5_dpp_ab.txt: up to 2^16 inputs the fold is latency-bound and sixteen times the lanes win
    // 1.56x; at 2^20 and above the scattered reads and the per-lane powers lose 2-4x)
    if (logf == 4 && (ctx->tune_fri_fold_lanes == 1 || (ctx->tune_fri_fold_lanes == 0 && logn <= 16))) {
        hipLaunchKernelGGL(fri_fold_lanes_kernel, dim3((unsigned)((m * 16 + 255) / 256)), block, 0, ctx->stream, a, (const u64 *)pl->d_tws);
        ZP_HIP(ctx, hipGetLastError());
        return ZP_OK;
    }
    switch (logf) {
        case 1: hipLaunchKernelGGL(fri_fold_kernel<1>, grid, block, 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(fri_fold_kernel<2>, grid, block, 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL(fri_fold_kernel<3>, grid, block, 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(fri_fold_kernel<4>, grid, block, 0, ctx->stream, a); break;
    }
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
