// BN254 (alt_bn128) G1 multi-scalar multiplication: sum_i s_i * P_i   (SURVEY.md 8a N6).
//
// No reference counterpart in /root/reference: the Groth16 wrap that answers GenFinalProof
// (proto/prover/v1/prover.proto:130-148, client src/prover/provider.rs:472-503) lives in the external
// prover.  Pippenger bucket method, window c bits:
//   1. digits     : counting sort of the point indices by window digit (histogram -> scan -> scatter)
//   2. bucket sums: one lane per (window, bucket) adds its points (Jacobian += affine, 7M+4S)
//   3. reduction  : running sums over segments of 64 buckets, segment weights by double-and-add,
//                   tree sum per window in LDS
//   4. the <= 32 window results are combined on the host (254 doublings).
// Field: F_q in Montgomery form, 8 x 32-bit limbs, CIOS with v_mad_u64_u32.  This first version is
// VALU only; the MFMA limb-product formulation north_star mentions is not built.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {

struct fq {
    u32 l[8];
};
#define FQ_HD __host__ __device__ __forceinline__

__device__ __constant__ const u32 FQ_Q_D[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                               0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
static const u32 FQ_Q_H[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                              0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
#define FQ_INV32 0xe4866389u

FQ_HD const u32 *fq_q() {
#if defined(__HIP_DEVICE_COMPILE__)
    return FQ_Q_D;
#else
    return FQ_Q_H;
#endif
}
FQ_HD fq fq_zero() {
    fq r;
    for (int i = 0; i < 8; i++) r.l[i] = 0;
    return r;
}
FQ_HD fq fq_one() {  // R mod q
    const u32 v[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    fq r;
    for (int i = 0; i < 8; i++) r.l[i] = v[i];
    return r;
}
FQ_HD fq fq_r2() {  // R^2 mod q
    const u32 v[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
    fq r;
    for (int i = 0; i < 8; i++) r.l[i] = v[i];
    return r;
}
FQ_HD bool fq_is_zero(const fq &a) {
    u32 o = 0;
    for (int i = 0; i < 8; i++) o |= a.l[i];
    return o == 0;
}
FQ_HD bool fq_eq(const fq &a, const fq &b) {
    u32 o = 0;
    for (int i = 0; i < 8; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
// r = t - q if (extra || t >= q) else t   -- branch-free (a divergent early-exit compare in every field
// operation serialises the wave)
FQ_HD fq fq_cond_sub(const u32 *t, u32 extra) {
    const u32 *q = fq_q();
    u32 d[8];
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 x = (u64)t[i] - q[i] - br;
        d[i] = (u32)x;
        br = (x >> 32) & 1;
    }
    const u32 use_d = 0u - (u32)((extra != 0) | (br == 0));   // all ones: take t - q
    fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = (d[i] & use_d) | (t[i] & ~use_d);
    return r;
}
FQ_HD fq fq_add(const fq &a, const fq &b) {
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)a.l[i] + b.l[i];
        t[i] = (u32)c;
        c >>= 32;
    }
    return fq_cond_sub(t, (u32)c);
}
FQ_HD fq fq_sub(const fq &a, const fq &b) {
    const u32 *q = fq_q();
    u32 t[8];
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const u64 d = (u64)a.l[i] - b.l[i] - br;
        t[i] = (u32)d;
        br = (d >> 32) & 1;
    }
    const u32 m = 0u - (u32)br;   // borrowed: add q back
    fq r;
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (u64)t[i] + (q[i] & m);
        r.l[i] = (u32)c;
        c >>= 32;
    }
    return r;
}
FQ_HD fq fq_dbl(const fq &a) { return fq_add(a, a); }
// Montgomery product a*b/R mod q (CIOS, 32-bit limbs)
FQ_HD fq fq_mul(const fq &a, const fq &b) {
    const u32 *q = fq_q();
    u32 t[10];
    for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u64 carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            u64 p = (u64)a.l[j] * b.l[i] + t[j] + carry;
            t[j] = (u32)p;
            carry = p >> 32;
        }
        u64 s = (u64)t[8] + carry;
        t[8] = (u32)s;
        t[9] = (u32)(s >> 32);
        const u32 m = t[0] * FQ_INV32;
        u64 p = (u64)m * q[0] + t[0];
        carry = p >> 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            p = (u64)m * q[j] + t[j] + carry;
            t[j - 1] = (u32)p;
            carry = p >> 32;
        }
        s = (u64)t[8] + carry;
        t[7] = (u32)s;
        t[8] = t[9] + (u32)(s >> 32);
    }
    return fq_cond_sub(t, t[8]);
}
FQ_HD fq fq_sqr(const fq &a) { return fq_mul(a, a); }
FQ_HD fq fq_to_mont(const fq &a) { return fq_mul(a, fq_r2()); }
FQ_HD fq fq_from_mont(const fq &a) {
    fq one = fq_zero();
    one.l[0] = 1;
    return fq_mul(a, one);
}

struct jac {
    fq X, Y, Z;  // Z == 0: point at infinity
};
FQ_HD jac jac_inf() {
    jac p;
    p.X = fq_one();
    p.Y = fq_one();
    p.Z = fq_zero();
    return p;
}
// dbl-2009-l (a = 0)
FQ_HD jac jac_dbl(const jac &p) {
    if (fq_is_zero(p.Z)) return p;
    fq A = fq_sqr(p.X), B = fq_sqr(p.Y), C = fq_sqr(B);
    fq t = fq_add(p.X, B);
    fq D = fq_dbl(fq_sub(fq_sub(fq_sqr(t), A), C));
    fq E = fq_add(fq_dbl(A), A);
    fq F = fq_sqr(E);
    jac r;
    r.X = fq_sub(F, fq_dbl(D));
    fq C8 = fq_dbl(fq_dbl(fq_dbl(C)));
    r.Y = fq_sub(fq_mul(E, fq_sub(D, r.X)), C8);
    r.Z = fq_dbl(fq_mul(p.Y, p.Z));
    return r;
}
// madd-2007-bl: Jacobian + affine (qx, qy) (affine point must not be infinity)
FQ_HD jac jac_madd(const jac &p, const fq &qx, const fq &qy) {
    if (fq_is_zero(p.Z)) {
        jac r;
        r.X = qx;
        r.Y = qy;
        r.Z = fq_one();
        return r;
    }
    fq Z1Z1 = fq_sqr(p.Z);
    fq U2 = fq_mul(qx, Z1Z1);
    fq S2 = fq_mul(fq_mul(qy, p.Z), Z1Z1);
    if (fq_eq(U2, p.X)) {
        if (fq_eq(S2, p.Y)) return jac_dbl(p);
        return jac_inf();
    }
    fq H = fq_sub(U2, p.X);
    fq HH = fq_sqr(H);
    fq I = fq_dbl(fq_dbl(HH));
    fq J = fq_mul(H, I);
    fq rr = fq_dbl(fq_sub(S2, p.Y));
    fq V = fq_mul(p.X, I);
    jac r;
    r.X = fq_sub(fq_sub(fq_sqr(rr), J), fq_dbl(V));
    r.Y = fq_sub(fq_mul(rr, fq_sub(V, r.X)), fq_dbl(fq_mul(p.Y, J)));
    r.Z = fq_sub(fq_sub(fq_sqr(fq_add(p.Z, H)), Z1Z1), HH);
    return r;
}
// add-2007-bl: Jacobian + Jacobian
FQ_HD jac jac_add(const jac &p, const jac &q) {
    if (fq_is_zero(p.Z)) return q;
    if (fq_is_zero(q.Z)) return p;
    fq Z1Z1 = fq_sqr(p.Z), Z2Z2 = fq_sqr(q.Z);
    fq U1 = fq_mul(p.X, Z2Z2), U2 = fq_mul(q.X, Z1Z1);
    fq S1 = fq_mul(fq_mul(p.Y, q.Z), Z2Z2), S2 = fq_mul(fq_mul(q.Y, p.Z), Z1Z1);
    if (fq_eq(U1, U2)) {
        if (fq_eq(S1, S2)) return jac_dbl(p);
        return jac_inf();
    }
    fq H = fq_sub(U2, U1);
    fq I = fq_sqr(fq_dbl(H));
    fq J = fq_mul(H, I);
    fq rr = fq_dbl(fq_sub(S2, S1));
    fq V = fq_mul(U1, I);
    jac r;
    r.X = fq_sub(fq_sub(fq_sqr(rr), J), fq_dbl(V));
    r.Y = fq_sub(fq_mul(rr, fq_sub(V, r.X)), fq_dbl(fq_mul(S1, J)));
    r.Z = fq_mul(fq_sub(fq_sub(fq_sqr(fq_add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
    return r;
}
FQ_HD jac jac_mul_small(const jac &p, u32 k) {  // k * p by double-and-add (k < 2^32)
    jac acc = jac_inf();
    for (int i = 31; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k >> i) & 1) acc = jac_add(acc, p);
    }
    return acc;
}

__device__ __forceinline__ u32 digit_of(const u32 *sc, int w, int c) {
    const int bit = w * c;
    const int limb = bit >> 5, off = bit & 31;
    u64 v = sc[limb];
    if (limb + 1 < 8) v |= (u64)sc[limb + 1] << 32;
    return (u32)(v >> off) & ((1u << c) - 1);
}

// ---- 1. counting sort of point indices by window digit
__global__ void __launch_bounds__(256) msm_hist_kernel(const u32 *scalars, u64 n, int c, int nwin, u32 *counts) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 sc[8];
    for (int k = 0; k < 8; k++) sc[k] = scalars[i * 8 + k];
    for (int w = 0; w < nwin; w++) {
        const u32 d = digit_of(sc, w, c);
        if (d) atomicAdd(&counts[((u64)w << c) + d], 1u);
    }
}
// exclusive scan of the 2^c counters of one window (one block per window)
__global__ void __launch_bounds__(1024) msm_scan_kernel(const u32 *counts, u32 *starts, u32 *cursor, int c) {
    __shared__ u32 part[1024];
    const u64 base = (u64)blockIdx.x << c;
    const u32 nb = 1u << c;
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per, hi = min(lo + per, nb);
    u32 s = 0;
    for (u32 b = lo; b < hi; b++) s += counts[base + b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int t = 0; t < 1024; t++) {
            u32 v = part[t];
            part[t] = run;
            run += v;
        }
    }
    __syncthreads();
    u32 run = part[threadIdx.x];
    for (u32 b = lo; b < hi; b++) {
        starts[base + b] = run;
        cursor[base + b] = run;
        run += counts[base + b];
    }
}
__global__ void __launch_bounds__(256) msm_scatter_kernel(const u32 *scalars, u64 n, int c, int nwin, u32 *cursor, u32 *sorted) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 sc[8];
    for (int k = 0; k < 8; k++) sc[k] = scalars[i * 8 + k];
    for (int w = 0; w < nwin; w++) {
        const u32 d = digit_of(sc, w, c);
        if (d) {
            const u32 pos = atomicAdd(&cursor[((u64)w << c) + d], 1u);
            sorted[(u64)w * n + pos] = (u32)i;
        }
    }
}
// ---- 2a. one-time conversion of the affine inputs to Montgomery form (16-byte vector accesses)
__global__ void __launch_bounds__(256) msm_to_mont_kernel(const uint4 *points, u64 n, uint4 *mont) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) q[k] = points[i * 4 + k];
    fq x, y;
    x.l[0] = q[0].x; x.l[1] = q[0].y; x.l[2] = q[0].z; x.l[3] = q[0].w; x.l[4] = q[1].x; x.l[5] = q[1].y; x.l[6] = q[1].z; x.l[7] = q[1].w;
    y.l[0] = q[2].x; y.l[1] = q[2].y; y.l[2] = q[2].z; y.l[3] = q[2].w; y.l[4] = q[3].x; y.l[5] = q[3].y; y.l[6] = q[3].z; y.l[7] = q[3].w;
    x = fq_to_mont(x);   // (0,0) stays (0,0): still the infinity marker
    y = fq_to_mont(y);
    mont[i * 4 + 0] = make_uint4(x.l[0], x.l[1], x.l[2], x.l[3]);
    mont[i * 4 + 1] = make_uint4(x.l[4], x.l[5], x.l[6], x.l[7]);
    mont[i * 4 + 2] = make_uint4(y.l[0], y.l[1], y.l[2], y.l[3]);
    mont[i * 4 + 3] = make_uint4(y.l[4], y.l[5], y.l[6], y.l[7]);
}
// ---- 2b. bucket sums: lane = (window, bucket); the next point is fetched while the current one is added
__global__ void __launch_bounds__(256) msm_bucket_kernel(const uint4 *mont, u64 n, int c, int nwin, const u32 *starts,
                                                        const u32 *counts, const u32 *sorted, jac *buckets) {
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    if (id >= ((u64)nwin << c)) return;
    const u64 w = id >> c;
    const u32 st = starts[id], cnt = counts[id];
    const u32 *idx = sorted + w * n + st;
    jac acc = jac_inf();
    uint4 nx[4];
    if (cnt) {
        const u64 p0 = idx[0];
#pragma unroll
        for (int k = 0; k < 4; k++) nx[k] = mont[p0 * 4 + k];
    }
    for (u32 k = 0; k < cnt; k++) {
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = nx[j];
        if (k + 1 < cnt) {
            const u64 pn = idx[k + 1];
#pragma unroll
            for (int j = 0; j < 4; j++) nx[j] = mont[pn * 4 + j];
        }
        fq x, y;
        x.l[0] = q[0].x; x.l[1] = q[0].y; x.l[2] = q[0].z; x.l[3] = q[0].w; x.l[4] = q[1].x; x.l[5] = q[1].y; x.l[6] = q[1].z; x.l[7] = q[1].w;
        y.l[0] = q[2].x; y.l[1] = q[2].y; y.l[2] = q[2].z; y.l[3] = q[2].w; y.l[4] = q[3].x; y.l[5] = q[3].y; y.l[6] = q[3].z; y.l[7] = q[3].w;
        if (fq_is_zero(x) && fq_is_zero(y)) continue;  // (0,0) encodes the point at infinity
        acc = jac_madd(acc, x, y);
    }
    buckets[id] = acc;
}
// ---- 3a. per segment of SEG buckets: sum_{b in seg} b * B_b
#define MSM_SEG 64
__global__ void __launch_bounds__(64) msm_segment_kernel(const jac *buckets, int c, int nwin, jac *segs) {
    const u64 id = (u64)blockIdx.x * 64 + threadIdx.x;
    const u64 segs_per_win = (1ULL << c) / MSM_SEG;
    if (id >= (u64)nwin * segs_per_win) return;
    const u64 w = id / segs_per_win, sidx = id % segs_per_win;
    const u64 s = sidx * MSM_SEG;
    const jac *B = buckets + (w << c) + s;
    jac run = jac_inf(), acc = jac_inf();
    for (int k = MSM_SEG - 1; k >= 1; k--) {
        run = jac_add(run, B[k]);
        acc = jac_add(acc, run);
    }
    run = jac_add(run, B[0]);                       // total of the segment
    if (s) acc = jac_add(acc, jac_mul_small(run, (u32)s));  // + s * total
    segs[id] = acc;
}
// ---- 3b. tree sum of the segment results of one window (one block per window)
__global__ void __launch_bounds__(256) msm_window_kernel(const jac *segs, int nseg, jac *wins) {
    __shared__ jac sh[256];
    const jac *S = segs + (u64)blockIdx.x * nseg;
    jac acc = jac_inf();
    for (int k = threadIdx.x; k < nseg; k += 256) acc = jac_add(acc, S[k]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = jac_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) wins[blockIdx.x] = sh[0];
}

fq fq_inv_host(const fq &a) {  // a^(q-2) in Montgomery form (host, once per MSM)
    u32 e[8];
    memcpy(e, FQ_Q_H, sizeof(e));
    e[0] -= 2;  // q is odd and its low limb is > 2
    fq r = fq_one(), b = a;
    for (int i = 0; i < 256; i++) {
        if ((e[i >> 5] >> (i & 31)) & 1) r = fq_mul(r, b);
        b = fq_sqr(b);
    }
    return r;
}

}  // namespace

extern "C" int32_t zp_msm_bn254(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n,
                                uint32_t *h_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "msm_bn254");
    ZP_ARG(ctx, h_out != nullptr, "null output");
    ZP_ARG(ctx, n < (1ULL << 31), "too many points");
    memset(h_out, 0, 16 * sizeof(uint32_t));
    if (n == 0) return ZP_OK;
    ZP_ARG(ctx, d_points && d_scalars, "null device pointer");
    int c = 4;
    while (c < 16 && (1ULL << (c + 2)) <= n) c++;   // ~4 points per bucket up to c = 16
    if (c < 6) c = 6;                                // segments of 64 buckets need c >= 6
    const int nwin = (254 + c - 1) / c;
    const u64 nb = (u64)nwin << c;
    u32 *d_counts = nullptr, *d_starts = nullptr, *d_cursor = nullptr, *d_sorted = nullptr;
    jac *d_buckets = nullptr, *d_segs = nullptr, *d_wins = nullptr;
    uint4 *d_mont = nullptr;
    const u64 nseg = (1ULL << c) / MSM_SEG;
    ZP_HIP(ctx, hipSetDevice(ctx->device));
    ZP_HIP(ctx, hipMalloc((void **)&d_counts, nb * 4 * 3));
    d_starts = d_counts + nb;
    d_cursor = d_starts + nb;
    ZP_HIP(ctx, hipMalloc((void **)&d_sorted, (u64)nwin * n * 4));
    ZP_HIP(ctx, hipMalloc((void **)&d_mont, (u64)n * 64));
    ZP_HIP(ctx, hipMalloc((void **)&d_buckets, (nb + nwin * nseg + nwin) * sizeof(jac)));
    d_segs = d_buckets + nb;
    d_wins = d_segs + nwin * nseg;
    ZP_HIP(ctx, hipMemsetAsync(d_counts, 0, nb * 4, ctx->stream));
    const unsigned gb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(msm_to_mont_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const uint4 *)d_points, (u64)n, d_mont);
    hipLaunchKernelGGL(msm_hist_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const u32 *)d_scalars, (u64)n, c, nwin, d_counts);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(nwin), dim3(1024), 0, ctx->stream, d_counts, d_starts, d_cursor, c);
    hipLaunchKernelGGL(msm_scatter_kernel, dim3(gb), dim3(256), 0, ctx->stream, (const u32 *)d_scalars, (u64)n, c, nwin, d_cursor, d_sorted);
    hipLaunchKernelGGL(msm_bucket_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint4 *)d_mont, (u64)n, c, nwin, d_starts, d_counts, d_sorted, d_buckets);
    hipLaunchKernelGGL(msm_segment_kernel, dim3((unsigned)((nwin * nseg + 63) / 64)), dim3(64), 0, ctx->stream, d_buckets, c, nwin, d_segs);
    hipLaunchKernelGGL(msm_window_kernel, dim3(nwin), dim3(256), 0, ctx->stream, d_segs, (int)nseg, d_wins);
    hipError_t le = hipGetLastError();
    std::vector<jac> wins(nwin);
    hipError_t ce = hipMemcpyAsync(wins.data(), d_wins, nwin * sizeof(jac), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t se = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_counts);
    (void)hipFree(d_sorted);
    (void)hipFree(d_mont);
    (void)hipFree(d_buckets);
    ZP_HIP(ctx, le);
    ZP_HIP(ctx, ce);
    ZP_HIP(ctx, se);
    // 4. host: result = sum_w 2^(c*w) * W_w  (Horner from the top window)
    jac acc = jac_inf();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) acc = jac_dbl(acc);
        acc = jac_add(acc, wins[w]);
    }
    if (fq_is_zero(acc.Z)) return ZP_OK;  // infinity: all-zero output
    fq zi = fq_inv_host(acc.Z);
    fq zi2 = fq_sqr(zi);
    fq x = fq_from_mont(fq_mul(acc.X, zi2));
    fq y = fq_from_mont(fq_mul(acc.Y, fq_mul(zi2, zi)));
    memcpy(h_out, x.l, 32);
    memcpy(h_out + 8, y.l, 32);
    return ZP_OK;
}
