// NTT over the BN254 scalar field F_r and the QAP quotient of a Groth16 prover (GenFinalProof, proto/prover/v1/prover.proto:130-148,
// src/prover/provider.rs:472-503: the final wrap; the reference holds none of the arithmetic).
//
// Transform: self-sorting Stockham passes of radix 2^L, L <= 8, natural order in and out, no bit-reversal pass.  With
// P = product of the radices of the earlier passes, a pass reads  x[u + (N/R) j]  (u = p P + q, q < P),  computes the
// R-point DFT over j in LDS (radix-2 stages, one butterfly per thread per stage), multiplies output k by the inter-pass
// twiddle w_N^(P p k) and writes  y[p P R + k P + q].  A tile is T consecutive u and all R rows: loads are runs of
// T x 32 B; stores are runs of T x 32 B (later passes) or whole R-element rows (first pass: the transposing copy-out).
//
// Arithmetic: fr254.hpp (nine 29-bit limbs, Montgomery product).  Data stays in STANDARD form in HBM: every multiplication
// is by a known constant, so all tables hold c * 2^261 (Montgomery form) and mont(x, c 2^261) = x c needs no conversion pass.
// The kernel is integer-VALU-bound by construction: ~6 field products per element per pass (L/2 in the stages, two for the
// two-level inter-pass twiddle), ~300 instructions each, against 64 B of HBM traffic.
#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "fr254.hpp"

namespace {

constexpr int FR_TILE = 1024;   // elements of one tile in LDS (36 KiB)

struct FrPassArgs {
    const u64 *src;
    u64 *dst;
    int logn, L, logP, logT, lb;
    int last;                 // no inter-pass twiddle (p = 0 everywhere)
    const u32 *twlo, *twhi;   // w^e = twlo[e & (2^lb - 1)] * twhi[e >> lb]
    const u32 *wr;            // w_R^e, e < R/2
    int in_mode;              // 1: multiply input i by  ilo[i & mask] * ihi[i >> lb]   (coset, forward)
    const u32 *ilo, *ihi;
    int out_mode;             // 1: multiply output by oconst;  2: by olo[k & mask] * ohi[k >> lb]
    const u32 *olo, *ohi;
    fr oconst;
};

__device__ __forceinline__ fr frl(const u32 *p) {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}
__device__ __forceinline__ fr lds_get(const u32 (*sh)[FR_TILE], int e) {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = sh[i][e];
    return r;
}
__device__ __forceinline__ void lds_put(u32 (*sh)[FR_TILE], int e, const fr &v) {
#pragma unroll
    for (int i = 0; i < 9; i++) sh[i][e] = v.l[i];
}

__global__ void __launch_bounds__(512) fr_ntt_pass_kernel(FrPassArgs a) {
    __shared__ u32 sh[9][FR_TILE];
    const int tid = threadIdx.x, NT = blockDim.x;          // NT = R T / 2
    const int R = 1 << a.L, T = 1 << a.logT;
    const u32 mask = (1u << a.lb) - 1;
    const size_t u0 = (size_t)blockIdx.x << a.logT;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const int e = tid + h * NT, j = e >> a.logT, t = e & (T - 1);
        const size_t idx = ((size_t)j << (a.logn - a.L)) + u0 + t;
        u64 w[4];
        const ulonglong2 w01 = *reinterpret_cast<const ulonglong2 *>(a.src + idx * 4);
        const ulonglong2 w23 = *reinterpret_cast<const ulonglong2 *>(a.src + idx * 4 + 2);
        w[0] = w01.x; w[1] = w01.y; w[2] = w23.x; w[3] = w23.y;
        fr x = fr_from_u64(w);
        if (a.in_mode == 1) {
            const fr g = fr_mul(frl(a.ilo + (size_t)((u32)idx & mask) * 9), frl(a.ihi + (idx >> a.lb) * 9));
            x = fr_mul(x, g);
        }
        lds_put(sh, e, x);
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < a.L; st++) {
        const int lh = a.L - 1 - st, half = 1 << lh;        // butterfly span
        const int t = tid & (T - 1), i = tid >> a.logT;     // i < R/2
        const int blk = i >> lh, off = i & (half - 1);
        const int eA = (((blk << (lh + 1)) + off) << a.logT) + t, eB = eA + (half << a.logT);
        const fr x = lds_get(sh, eA), y = lds_get(sh, eB);
        const fr s = fr_add(x, y);
        fr d = fr_sub(x, y);
        if (half > 1) d = fr_mul(d, frl(a.wr + (size_t)(off << st) * 9));
        lds_put(sh, eA, s);
        lds_put(sh, eB, d);
        __syncthreads();
    }
    // position pos of the tile now holds output k = bitrev_L(pos)
    const int logP = a.logP;
#pragma unroll 1
    for (int h = 0; h < 2; h++) {
        const int e = tid + h * NT;
        int t, k;
        size_t oidx, p;
        if (logP == 0) {              // first pass: k fastest, whole rows of R outputs per column
            t = e >> a.L; k = e & (R - 1);
            p = u0 + t;
            oidx = (p << a.L) + k;
        } else {                      // T <= P: the tile shares p
            t = e & (T - 1);
            k = (int)(__brev((u32)(e >> a.logT)) >> (32 - a.L));
            p = u0 >> logP;
            const size_t q = (u0 & (((size_t)1 << logP) - 1)) + t;
            oidx = (((p << a.L) + k) << logP) + q;
        }
        const int pos = logP == 0 ? (int)(__brev((u32)k) >> (32 - a.L)) : (e >> a.logT);
        fr x = lds_get(sh, (pos << a.logT) + t);
        if (!a.last) {
            const size_t ex = (p * (size_t)k) << logP;      // < N
            const fr tw = fr_mul(frl(a.twlo + (size_t)((u32)ex & mask) * 9), frl(a.twhi + (ex >> a.lb) * 9));
            x = fr_mul(x, tw);
        }
        if (a.out_mode == 1) {
            x = fr_mul(x, a.oconst);
        } else if (a.out_mode == 2) {
            const fr g = fr_mul(frl(a.olo + (size_t)((u32)oidx & mask) * 9), frl(a.ohi + (oidx >> a.lb) * 9));
            x = fr_mul(x, g);
        }
        u64 w[4];
        fr_to_u64(x, w);
        *reinterpret_cast<ulonglong2 *>(a.dst + oidx * 4) = make_ulonglong2(w[0], w[1]);
        *reinterpret_cast<ulonglong2 *>(a.dst + oidx * 4 + 2) = make_ulonglong2(w[2], w[3]);
    }
}

// a[i] = (a[i] b[i] - c[i]) s      s1 = s 2^522 mod r, s2 = s 2^261 mod r (plain values): mont(mont(a,b), s1) = a b s, mont(c, s2) = c s
__global__ void __launch_bounds__(256) fr_qap_pointwise_kernel(u64 *a, const u64 *b, const u64 *c, size_t n, fr s1, fr s2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 wa[4], wb[4], wc[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { wa[k] = a[i * 4 + k]; wb[k] = b[i * 4 + k]; wc[k] = c[i * 4 + k]; }
    const fr ab = fr_mul(fr_mul(fr_from_u64(wa), fr_from_u64(wb)), s1);
    const fr cs = fr_mul(fr_from_u64(wc), s2);
    fr_to_u64(fr_sub(ab, cs), wa);
#pragma unroll
    for (int k = 0; k < 4; k++) a[i * 4 + k] = wa[k];
}

// ---- host-side field helpers (fr254.hpp is host/device) ----
fr h_from_words(const u64 *w) { return fr_to_mont(fr_from_u64(w)); }               // -> Montgomery
fr h_pow(fr base_m, const u64 *ex, int nwords) {                                    // Montgomery in, Montgomery out
    fr r = fr_one();
    for (int k = nwords - 1; k >= 0; k--)
        for (int b = 63; b >= 0; b--) {
            r = fr_mul(r, r);
            if ((ex[k] >> b) & 1) r = fr_mul(r, base_m);
        }
    return r;
}
fr h_inv(fr a_m) {                                                                  // a^(r-2)
    const u64 e[4] = {0x43e1f593f0000001ULL - 2, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    return h_pow(a_m, e, 4);
}
fr h_pow_u64(fr base_m, u64 e) { return h_pow(base_m, &e, 1); }
fr h_root(int logn) {                                                               // 5^((r-1)/2^logn), Montgomery
    // (r - 1) >> 28
    const u64 r1[4] = {0x43e1f593f0000000ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    u64 e[4];
    for (int k = 0; k < 4; k++) e[k] = (r1[k] >> 28) | (k + 1 < 4 ? r1[k + 1] << 36 : 0);
    const u64 five[4] = {5, 0, 0, 0};
    fr w = h_pow(h_from_words(five), e, 4);                                          // primitive 2^28-th root
    for (int i = logn; i < 28; i++) w = fr_mul(w, w);
    return w;
}

struct FrTables {
    int logn = 0, lb = 0;
    u32 *d_lo = nullptr, *d_hi = nullptr;   // two-level powers of `base`, `pre` folded into lo
};
// two-level power table of base (Montgomery): lo[e] = pre * base^e (e < 2^lb), hi[e] = base^(e 2^lb)
int32_t build_two_level(zp_ctx *ctx, fr base, fr pre, int logn, FrTables *out) {
    const int lb = (logn + 1) / 2;
    const size_t nlo = (size_t)1 << lb, nhi = (size_t)1 << (logn - lb);
    std::vector<u32> lo(nlo * 9), hi(nhi * 9);
    fr v = pre;
    for (size_t e = 0; e < nlo; e++) { memcpy(&lo[e * 9], v.l, 36); v = fr_mul(v, base); }
    fr step = base;
    for (int i = 0; i < lb; i++) step = fr_mul(step, step);
    v = fr_one();
    for (size_t e = 0; e < nhi; e++) { memcpy(&hi[e * 9], v.l, 36); v = fr_mul(v, step); }
    out->logn = logn; out->lb = lb;
    ZP_HIP(ctx, hipMalloc(&out->d_lo, lo.size() * 4));
    ZP_HIP(ctx, hipMalloc(&out->d_hi, hi.size() * 4));
    ZP_HIP(ctx, hipMemcpy(out->d_lo, lo.data(), lo.size() * 4, hipMemcpyHostToDevice));
    ZP_HIP(ctx, hipMemcpy(out->d_hi, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    return ZP_OK;
}

struct FrPlan {
    int logn = 0, npass = 0, L[4] = {0, 0, 0, 0};
    FrTables tw;            // powers of w (direction matched)
    u32 *d_wr[4] = {nullptr, nullptr, nullptr, nullptr};   // per pass: w_R^e, e < R/2
    fr ninv;                // 1/N, Montgomery
};
std::mutex g_mu;
std::map<std::string, FrPlan> g_plans;       // key: device, logn, direction
std::map<std::string, FrTables> g_cosets;    // key: device, logn, direction, coset words

std::string key_of(int dev, int logn, int inv, const u64 *cw) {
    char buf[160];
    snprintf(buf, sizeof buf, "%d/%d/%d/%016llx%016llx%016llx%016llx", dev, logn, inv, cw ? (unsigned long long)cw[3] : 0ULL,
             cw ? (unsigned long long)cw[2] : 0ULL, cw ? (unsigned long long)cw[1] : 0ULL, cw ? (unsigned long long)cw[0] : 0ULL);
    return buf;
}

int32_t get_plan(zp_ctx *ctx, int logn, bool inverse, FrPlan **out) {
    const std::string key = key_of(ctx->device, logn, inverse, nullptr);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { *out = &it->second; return ZP_OK; }
    FrPlan pl;
    pl.logn = logn;
    pl.npass = (logn + 7) / 8;
    for (int i = 0; i < pl.npass; i++) pl.L[i] = logn / pl.npass + (i < logn % pl.npass ? 1 : 0);
    fr w = h_root(logn);
    if (inverse) w = h_inv(w);
    ZP_TRY(build_two_level(ctx, w, fr_one(), logn, &pl.tw));
    for (int i = 0; i < pl.npass; i++) {
        const int R = 1 << pl.L[i];
        fr wR = w;                                           // w^(N/R)
        for (int s = 0; s < logn - pl.L[i]; s++) wR = fr_mul(wR, wR);
        std::vector<u32> t((size_t)(R / 2 > 0 ? R / 2 : 1) * 9);
        fr v = fr_one();
        for (int e = 0; e < R / 2; e++) { memcpy(&t[(size_t)e * 9], v.l, 36); v = fr_mul(v, wR); }
        ZP_HIP(ctx, hipMalloc(&pl.d_wr[i], t.size() * 4));
        ZP_HIP(ctx, hipMemcpy(pl.d_wr[i], t.data(), t.size() * 4, hipMemcpyHostToDevice));
    }
    u64 nn[4] = {(u64)1 << logn, 0, 0, 0};
    pl.ninv = h_inv(h_from_words(nn));
    *out = &(g_plans[key] = pl);
    return ZP_OK;
}

int32_t get_coset(zp_ctx *ctx, int logn, bool inverse, const u64 *cw, const fr &ninv, FrTables **out) {
    const std::string key = key_of(ctx->device, logn, inverse, cw);
    auto it = g_cosets.find(key);
    if (it != g_cosets.end()) { *out = &it->second; return ZP_OK; }
    fr g = h_from_words(cw);
    FrTables t;
    if (inverse) ZP_TRY(build_two_level(ctx, h_inv(g), ninv, logn, &t));   // g^-k / N
    else ZP_TRY(build_two_level(ctx, g, fr_one(), logn, &t));              // g^i
    *out = &(g_cosets[key] = t);
    return ZP_OK;
}

int32_t run_ntt(zp_ctx *ctx, u64 *d_data, int logn, bool inverse, const u64 *h_coset) {
    if (logn == 0) return ZP_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    FrPlan *pl;
    ZP_TRY(get_plan(ctx, logn, inverse, &pl));
    FrTables *cs = nullptr;
    if (h_coset) ZP_TRY(get_coset(ctx, logn, inverse, h_coset, pl->ninv, &cs));
    const size_t N = (size_t)1 << logn;
    u64 *tmp;
    ZP_TRY(zpi_scratch(ctx, 0, N * 4, &tmp));
    const u64 *src = d_data;
    u64 *dst = tmp;
    int logP = 0;
    for (int i = 0; i < pl->npass; i++) {
        FrPassArgs a;
        memset(&a, 0, sizeof a);
        a.src = src; a.dst = dst; a.logn = logn; a.L = pl->L[i]; a.logP = logP; a.lb = pl->tw.lb;
        a.last = i == pl->npass - 1;
        int logT = 10 - a.L;                                   // R T = 1024
        if (logT > logn - a.L) logT = logn - a.L;              // T <= N / R
        if (i > 0 && logT > logP) logT = logP;                 // T <= P: the tile shares p
        a.logT = logT;
        a.twlo = pl->tw.d_lo; a.twhi = pl->tw.d_hi; a.wr = pl->d_wr[i];
        if (i == 0 && cs && !inverse) { a.in_mode = 1; a.ilo = cs->d_lo; a.ihi = cs->d_hi; }
        if (a.last && inverse) {
            if (cs) { a.out_mode = 2; a.olo = cs->d_lo; a.ohi = cs->d_hi; }
            else { a.out_mode = 1; a.oconst = pl->ninv; }
        }
        const int threads = 1 << (a.L + logT - 1);
        const size_t tiles = N >> (a.L + logT);
        ZP_ARG(ctx, threads >= 1 && threads <= 512 && tiles > 0 && tiles < (1u << 31), "F_r NTT launch shape");
        hipLaunchKernelGGL(fr_ntt_pass_kernel, dim3((unsigned)tiles), dim3(threads), 0, ctx->stream, a);
        ZP_HIP(ctx, hipGetLastError());
        logP += a.L;
        src = dst;
        dst = (dst == tmp) ? d_data : tmp;
    }
    if (src != d_data) ZP_HIP(ctx, hipMemcpyAsync(d_data, src, N * 32, hipMemcpyDeviceToDevice, ctx->stream));
    return ZP_OK;
}

}  // namespace

extern "C" {

int32_t zp_ntt_bn254(zp_ctx *ctx, uint64_t *d_data_, int32_t logn, int32_t inverse, const uint64_t *h_coset_) {
    if (!ctx) return ZP_ERR_ARG;
    u64 *d_data = reinterpret_cast<u64 *>(d_data_);
    const u64 *h_coset = reinterpret_cast<const u64 *>(h_coset_);
    ZpStage stage(ctx, "ntt_bn254");
    ZP_ARG(ctx, d_data != nullptr, "null buffer");
    ZP_ARG(ctx, logn >= 0 && logn <= 28, "logn must be in [0, 28] (2-adicity of r - 1)");
    if (h_coset) {
        ZP_ARG(ctx, fr_is_canonical_u64(h_coset), "coset shift must be < r");
        ZP_ARG(ctx, (h_coset[0] | h_coset[1] | h_coset[2] | h_coset[3]) != 0, "coset shift must be non-zero");
    }
    return run_ntt(ctx, d_data, logn, inverse != 0, h_coset);
}

int32_t zp_qap_quotient_bn254(zp_ctx *ctx, uint64_t *d_a_, uint64_t *d_b_, uint64_t *d_c_, int32_t logm, const uint64_t *h_coset_) {
    if (!ctx) return ZP_ERR_ARG;
    u64 *d_a = reinterpret_cast<u64 *>(d_a_), *d_b = reinterpret_cast<u64 *>(d_b_), *d_c = reinterpret_cast<u64 *>(d_c_);
    const u64 *h_coset = reinterpret_cast<const u64 *>(h_coset_);
    ZpStage stage(ctx, "qap_quotient_bn254");
    ZP_ARG(ctx, d_a && d_b && d_c && h_coset, "null buffer");
    ZP_ARG(ctx, logm >= 1 && logm <= 28, "logm must be in [1, 28]");
    ZP_ARG(ctx, fr_is_canonical_u64(h_coset) && (h_coset[0] | h_coset[1] | h_coset[2] | h_coset[3]) != 0, "coset shift must be in [1, r)");
    const size_t m = (size_t)1 << logm;
    // Z(x) = x^m - 1 is the constant g^m - 1 on the coset g <w>; it must not vanish
    const fr g = h_from_words(h_coset);
    fr z = fr_sub(h_pow_u64(g, (u64)m), fr_one());
    u64 zw[4];
    fr_to_u64(fr_from_mont(z), zw);
    ZP_ARG(ctx, (zw[0] | zw[1] | zw[2] | zw[3]) != 0, "coset shift lies in the evaluation domain");
    const fr zinv_m = h_inv(z);                         // s 2^261
    const fr s2 = zinv_m;                               // plain value s 2^261: exactly the limbs of the Montgomery form
    const fr s1 = fr_mul(zinv_m, fr_r2());              // s 2^522 = mont(s 2^261, 2^522)
    for (u64 *d : {d_a, d_b, d_c}) {
        ZP_TRY(run_ntt(ctx, d, logm, true, nullptr));
        ZP_TRY(run_ntt(ctx, d, logm, false, h_coset));
    }
    hipLaunchKernelGGL(fr_qap_pointwise_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, d_a, d_b, d_c, m, s1, s2);
    ZP_HIP(ctx, hipGetLastError());
    return run_ntt(ctx, d_a, logm, true, h_coset);
}

}  // extern "C"
