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
    switch (logf) {
        case 1: hipLaunchKernelGGL(fri_fold_kernel<1>, grid, block, 0, ctx->stream, a); break;
        case 2: hipLaunchKernelGGL(fri_fold_kernel<2>, grid, block, 0, ctx->stream, a); break;
        case 3: hipLaunchKernelGGL(fri_fold_kernel<3>, grid, block, 0, ctx->stream, a); break;
        default: hipLaunchKernelGGL(fri_fold_kernel<4>, grid, block, 0, ctx->stream, a); break;
    }
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
