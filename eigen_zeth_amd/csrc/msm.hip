// BN254 (alt_bn128) G1 multi-scalar multiplication: sum_i s_i * P_i   (SURVEY.md 8a N6).
//
// No reference counterpart in /root/reference: the Groth16 wrap that answers GenFinalProof
// (proto/prover/v1/prover.proto:130-148, client src/prover/provider.rs:472-503) lives in the external
// prover.  Pippenger bucket method, window c bits:
//   1. digits     : counting sort of the point indices by window digit (histogram -> scan -> scatter)
//   2. bucket sums: one lane per (window, bucket) adds its points (Jacobian += affine, 7M+4S)
//   3. reduction  : running sums over segments of MSM_SEG buckets, segment weights by double-and-add,
//                   tree sum per window in LDS
//   4. the <= 32 window results are combined on the host (254 doublings).
// Field: F_q in Montgomery form, 9 x 29-bit limbs, product scanning with v_mad_u64_u32 (see below).  VALU only;
// the MFMA limb-product formulation north_star mentions is not built.
#include <hip/hip_runtime.h>

#include <type_traits>

#include <cstring>
#include <vector>

#include <thread>

#include "ctx.hpp"

namespace {

// ---- F_q in Montgomery form with R = 2^261: NINE 29-BIT LIMBS.
// v_mad_u64_u32 has a carry-out but no carry-in, so a 32-bit-limb multiplier spends two thirds of its instructions
// moving carries around (the first version compiled to ~790 instructions per product, 47 % of them v_mov).  With
// 29-bit limbs a 64-bit column accumulator takes all 18 products of a column (18 * 2^58 < 2^63) without any carry
// handling: one v_mad_u64_u32 per product, one shift per column -- 162 mads + ~110 other instructions.
// Values are always fully reduced (< q) and normalised (limbs < 2^29) between operations, so equality and the
// infinity tests are plain limb comparisons.
#define FQ_B 29
#define FQ_MASK 0x1FFFFFFFu
#define FQ_INV29 0x04866389u   // -q^-1 mod 2^29
struct fq {
    u32 l[9];
};
#define FQ_HD __host__ __device__ __forceinline__

#define FQ_Q0 0x187cfd47u
#define FQ_Q1 0x010460b6u
#define FQ_Q2 0x1c72a34fu
#define FQ_Q3 0x02d522d0u
#define FQ_Q4 0x1585d978u
#define FQ_Q5 0x02db40c0u
#define FQ_Q6 0x00a6e141u
#define FQ_Q7 0x0e5c2634u
#define FQ_Q8 0x0030644eu
// q as 8 x 32-bit words (exponent bits of the host inversion)
static const u32 FQ_Q_H[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                              0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};

FQ_HD u32 fq_q(int i) {
    switch (i) {
        case 0: return FQ_Q0;
        case 1: return FQ_Q1;
        case 2: return FQ_Q2;
        case 3: return FQ_Q3;
        case 4: return FQ_Q4;
        case 5: return FQ_Q5;
        case 6: return FQ_Q6;
        case 7: return FQ_Q7;
        default: return FQ_Q8;
    }
}
FQ_HD fq fq_zero() {
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
FQ_HD fq fq_one() {  // R mod q
    const u32 v[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = v[i];
    return r;
}
FQ_HD fq fq_r2() {  // R^2 mod q
    const u32 v[9] = {0x059bac10u, 0x0d1503a3u, 0x018016b8u, 0x10ab0ca8u, 0x02632639u, 0x02c0169fu, 0x169bfd53u, 0x11869d4cu, 0x002a11a6u};
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = v[i];
    return r;
}
FQ_HD bool fq_is_zero(const fq &a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}
FQ_HD bool fq_eq(const fq &a, const fq &b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) o |= a.l[i] ^ b.l[i];
    return o == 0;
}
// t: limbs possibly unnormalised (each < 2^31), value < 2q  ->  normalised value mod q.  Branch-free: both
// the carry-propagated t and t - q are formed, the sign of the last borrow selects.
FQ_HD fq fq_norm_sub(const u32 *t) {
    u32 n[9], d[9];
    int cn = 0, cd = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int vn = (int)t[i] + cn;
        n[i] = (u32)vn & FQ_MASK;
        cn = vn >> FQ_B;
        const int vd = (int)t[i] - (int)fq_q(i) + cd;
        d[i] = (u32)vd & FQ_MASK;
        cd = vd >> FQ_B;   // arithmetic shift: -1 on borrow
    }
    const u32 use_d = cd < 0 ? 0u : 0xFFFFFFFFu;   // no final borrow: t >= q
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (d[i] & use_d) | (n[i] & ~use_d);
    return r;
}
FQ_HD fq fq_add(const fq &a, const fq &b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = a.l[i] + b.l[i];
    return fq_norm_sub(t);
}
FQ_HD fq fq_sub(const fq &a, const fq &b) {
    // a - b, plus q when negative: both chains, select by the final borrow of a - b
    u32 d[9], e[9];
    int cd = 0, ce = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int vd = (int)a.l[i] - (int)b.l[i] + cd;
        d[i] = (u32)vd & FQ_MASK;
        cd = vd >> FQ_B;
        const int ve = (int)a.l[i] - (int)b.l[i] + (int)fq_q(i) + ce;
        e[i] = (u32)ve & FQ_MASK;
        ce = ve >> FQ_B;
    }
    const u32 use_e = cd < 0 ? 0xFFFFFFFFu : 0u;
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (e[i] & use_e) | (d[i] & ~use_e);
    return r;
}
FQ_HD fq fq_dbl(const fq &a) { return fq_add(a, a); }
// Montgomery product a*b/R mod q, R = 2^261: product scanning, one 64-bit accumulator per column
FQ_HD fq fq_mul(const fq &a, const fq &b) {
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fq_q(k - i);
        m[k] = ((u32)acc * FQ_INV29) & FQ_MASK;
        acc += (u64)m[k] * FQ_Q0;
        acc >>= FQ_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (u64)a.l[i] * b.l[k - i];
            acc += (u64)m[i] * fq_q(k - i);
        }
        t[k - 9] = (u32)acc & FQ_MASK;
        acc >>= FQ_B;
    }
    t[8] = (u32)acc;
    return fq_norm_sub(t);   // (ab + mq)/R < q (q/R + 1) < 2q
}
FQ_HD fq fq_sqr(const fq &a) { return fq_mul(a, a); }
FQ_HD fq fq_to_mont(const fq &a) { return fq_mul(a, fq_r2()); }
FQ_HD fq fq_from_mont(const fq &a) {
    fq one = fq_zero();
    one.l[0] = 1;
    return fq_mul(a, one);
}
// 8 x 32-bit words (little endian, value < 2^256... here always < q) <-> 9 x 29-bit limbs
FQ_HD fq fq_from_words(const u32 *w) {
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = FQ_B * i, k = bit >> 5, off = bit & 31;
        u64 v = w[k];
        if (k + 1 < 8) v |= (u64)w[k + 1] << 32;
        r.l[i] = (u32)(v >> off) & FQ_MASK;
    }
    return r;
}
FQ_HD void fq_to_words(const fq &a, u32 *w) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        // word k holds bits [32k, 32k+32): limbs i with 29i < 32k+32 and 29i+29 > 32k
        u64 v = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int lo = FQ_B * i - 32 * k;   // position of limb i relative to word k
            if (lo > -FQ_B && lo < 32) v |= lo >= 0 ? ((u64)a.l[i] << lo) : ((u64)a.l[i] >> (-lo));
        }
        w[k] = (u32)v;
    }
}

// ---- F_q2 = F_q[u]/(u^2 + 1)  (G2 coordinates)
struct fq2 {
    fq c0, c1;
};
FQ_HD fq2 fq2_make(const fq &a, const fq &b) {
    fq2 r;
    r.c0 = a;
    r.c1 = b;
    return r;
}

// ---- lazy (unreduced) arithmetic for the G1 bucket sums.  A point addition is eleven products with a dozen additions and
// subtractions between them; in the canonical form above every one of those normalises and conditionally subtracts q (two
// carry chains and a select, ~90 instructions -- half the instructions of a point addition).  R = 2^261 is 169 q, so a
// Montgomery product only needs  a b < 169 q^2  to return a value < 2 q, and the 64-bit column accumulators take limbs up to
// 2^30.  Between the products values therefore stay congruent but unreduced:
//   N(k): limbs 0..7 < 2^29, value < k q (what lz_mul, lz_sub, lz_carry return);   W(k): limbs < 2^30 (lz_add / lz_dbl of N values)
//   lz_sub<K>(a, b) = a - b + K q with ONE signed carry pass (K q >= b keeps it non-negative): ~45 two-cycle instructions.
// The bounds of every step of the mixed addition are in jac_madd_lazy; results are made canonical once, when a bucket is stored.
constexpr u32 FQ_QL[9] = {FQ_Q0, FQ_Q1, FQ_Q2, FQ_Q3, FQ_Q4, FQ_Q5, FQ_Q6, FQ_Q7, FQ_Q8};
constexpr u32 fq_kq_limb(int K, int i) {            // limb i of K q (normalised limbs, the top one takes the rest)
    u64 carry = 0, v = 0;
    for (int j = 0; j <= i; j++) {
        v = (u64)FQ_QL[j] * (u64)K + carry;
        carry = v >> FQ_B;
    }
    return i == 8 ? (u32)v : (u32)v & FQ_MASK;
}
__device__ __forceinline__ fq lz_mul(const fq &a, const fq &b) {      // limbs < 2^30, a b < 169 q^2  ->  N(1 + a b / 169 q^2)
    u32 m[9];
    fq r;
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fq_q(k - i);
        m[k] = ((u32)acc * FQ_INV29) & FQ_MASK;
        acc += (u64)m[k] * FQ_Q0;
        acc >>= FQ_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (u64)a.l[i] * b.l[k - i];
            acc += (u64)m[i] * fq_q(k - i);
        }
        r.l[k - 9] = (u32)acc & FQ_MASK;
        acc >>= FQ_B;
    }
    r.l[8] = (u32)acc;
    return r;
}
__device__ __forceinline__ fq lz_add(const fq &a, const fq &b) {      // N + N -> W
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
__device__ __forceinline__ fq lz_dbl(const fq &a) {                   // N -> W
    fq r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] << 1;
    return r;
}
__device__ __forceinline__ fq lz_quad(const fq &a) {                  // 4 a, carried: N -> N
    fq r;
    u32 c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u32 v = (a.l[i] << 2) + c;
        r.l[i] = i < 8 ? (v & FQ_MASK) : v;
        c = v >> FQ_B;
    }
    return r;
}
template <int K>
__device__ __forceinline__ fq lz_sub(const fq &a, const fq &b) {      // a - b + K q  (K q >= b):  -> N(a + K)
    fq r;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int v = (int)a.l[i] - (int)b.l[i] + (int)fq_kq_limb(K, i) + c;
        r.l[i] = i < 8 ? ((u32)v & FQ_MASK) : (u32)v;
        c = v >> FQ_B;        // arithmetic shift: floor division
    }
    return r;
}
template <int K>
__device__ __forceinline__ fq lz_sub2(const fq &a, const fq &b, const fq &d) {   // a - b - d + K q  (K q >= b + d)
    fq r;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int v = (int)a.l[i] - (int)b.l[i] - (int)d.l[i] + (int)fq_kq_limb(K, i) + c;
        r.l[i] = i < 8 ? ((u32)v & FQ_MASK) : (u32)v;
        c = v >> FQ_B;
    }
    return r;
}
__device__ __forceinline__ bool lz_is_zero_mod_q(const fq &a) {       // a in N(2): congruent to 0 iff a is 0 or q
    u32 z = 0, e = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) { z |= a.l[i]; e |= a.l[i] ^ FQ_QL[i]; }
    return z == 0 || e == 0;
}
__device__ __forceinline__ fq lz_canon(const fq &a) { return fq_mul(a, fq_one()); }   // a R / R mod q, fully reduced (limbs < 2^30, a < 169 q)

// ---- one set of names over both fields, so that the curve code and the kernels are written once
template <class F> struct FT;
template <> struct FT<fq> {
    static constexpr int WORDS = 8;   // 32-bit words of one packed element
    static FQ_HD fq zero() { return fq_zero(); }
    static FQ_HD fq one() { return fq_one(); }
    static FQ_HD fq from_words(const u32 *w) { return fq_from_words(w); }
    static FQ_HD void to_words(const fq &a, u32 *w) { fq_to_words(a, w); }
};
template <> struct FT<fq2> {
    static constexpr int WORDS = 16;
    static FQ_HD fq2 zero() { return fq2_make(fq_zero(), fq_zero()); }
    static FQ_HD fq2 one() { return fq2_make(fq_one(), fq_zero()); }
    static FQ_HD fq2 from_words(const u32 *w) { return fq2_make(fq_from_words(w), fq_from_words(w + 8)); }
    static FQ_HD void to_words(const fq2 &a, u32 *w) {
        fq_to_words(a.c0, w);
        fq_to_words(a.c1, w + 8);
    }
};
FQ_HD bool f_is_zero(const fq &a) { return fq_is_zero(a); }
FQ_HD bool f_eq(const fq &a, const fq &b) { return fq_eq(a, b); }
FQ_HD fq f_add(const fq &a, const fq &b) { return fq_add(a, b); }
FQ_HD fq f_sub(const fq &a, const fq &b) { return fq_sub(a, b); }
FQ_HD fq f_dbl(const fq &a) { return fq_dbl(a); }
FQ_HD fq f_mul(const fq &a, const fq &b) { return fq_mul(a, b); }
FQ_HD fq f_sqr(const fq &a) { return fq_sqr(a); }
FQ_HD fq f_to_mont(const fq &a) { return fq_to_mont(a); }
FQ_HD fq f_from_mont(const fq &a) { return fq_from_mont(a); }
FQ_HD bool f_is_zero(const fq2 &a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
FQ_HD bool f_eq(const fq2 &a, const fq2 &b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }
FQ_HD fq2 f_add(const fq2 &a, const fq2 &b) { return fq2_make(fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)); }
FQ_HD fq2 f_sub(const fq2 &a, const fq2 &b) { return fq2_make(fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)); }
FQ_HD fq2 f_dbl(const fq2 &a) { return fq2_make(fq_dbl(a.c0), fq_dbl(a.c1)); }
// (a b + c d) / R mod q with ONE Montgomery reduction: the two products share the column accumulators (27 terms of < 2^58 per
// column stay below 2^64).  Limbs < 2^29 (one operand of each product may have limbs < 2^30), a b + c d < 169 q^2.
FQ_HD fq fq_mul2(const fq &a, const fq &b, const fq &c, const fq &d) {
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (u64)a.l[i] * b.l[k - i];
            acc += (u64)c.l[i] * d.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fq_q(k - i);
        m[k] = ((u32)acc * FQ_INV29) & FQ_MASK;
        acc += (u64)m[k] * FQ_Q0;
        acc >>= FQ_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (u64)a.l[i] * b.l[k - i];
            acc += (u64)c.l[i] * d.l[k - i];
            acc += (u64)m[i] * fq_q(k - i);
        }
        t[k - 9] = (u32)acc & FQ_MASK;
        acc >>= FQ_B;
    }
    t[8] = (u32)acc;
    return fq_norm_sub(t);   // (ab + cd + mq)/R < q (2q/R + 1) < 2q
}
// q - a in (0, q] with one signed carry pass (a canonical); congruent to -a, limbs < 2^29
FQ_HD fq fq_neg_lazy(const fq &a) {
    fq r;
    int c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int v = (int)fq_q(i) - (int)a.l[i] + c;
        r.l[i] = i < 8 ? ((u32)v & FQ_MASK) : (u32)v;
        c = v >> FQ_B;
    }
    return r;
}
// F_q2 products with the reduction delayed over the two terms of each component (no Karatsuba: the same 486 limb products,
// two Montgomery reductions instead of three and none of its five additions / subtractions):
//   (a0 + a1 u)(b0 + b1 u) = (a0 b0 + (q - a1) b1) + (a0 b1 + a1 b0) u
FQ_HD fq2 f_mul(const fq2 &a, const fq2 &b) {
    return fq2_make(fq_mul2(a.c0, b.c0, fq_neg_lazy(a.c1), b.c1), fq_mul2(a.c0, b.c1, a.c1, b.c0));
}
FQ_HD fq2 f_sqr(const fq2 &a) {   // (a0 a0 + (q - a1) a1) + (a0 * 2 a1) u
    fq d;
#pragma unroll
    for (int i = 0; i < 9; i++) d.l[i] = a.c1.l[i] << 1;      // 2 a1, limbs < 2^30
    return fq2_make(fq_mul2(a.c0, a.c0, fq_neg_lazy(a.c1), a.c1), fq_mul(a.c0, d));
}
FQ_HD fq2 f_to_mont(const fq2 &a) { return fq2_make(fq_to_mont(a.c0), fq_to_mont(a.c1)); }
FQ_HD fq2 f_from_mont(const fq2 &a) { return fq2_make(fq_from_mont(a.c0), fq_from_mont(a.c1)); }

// ---- short Weierstrass curve y^2 = x^3 + b (a = 0) in Jacobian coordinates over F (G1: F_q, G2: F_q2)
template <class F>
struct jacT {
    F X, Y, Z;  // Z == 0: point at infinity
};
template <class F>
FQ_HD jacT<F> jac_inf() {
    jacT<F> p;
    p.X = FT<F>::one();
    p.Y = FT<F>::one();
    p.Z = FT<F>::zero();
    return p;
}
// dbl-2009-l (a = 0)
template <class F>
FQ_HD jacT<F> jac_dbl(const jacT<F> &p) {
    if (f_is_zero(p.Z)) return p;
    F A = f_sqr(p.X), B = f_sqr(p.Y), C = f_sqr(B);
    F t = f_add(p.X, B);
    F D = f_dbl(f_sub(f_sub(f_sqr(t), A), C));
    F E = f_add(f_dbl(A), A);
    F G = f_sqr(E);
    jacT<F> r;
    r.X = f_sub(G, f_dbl(D));
    F C8 = f_dbl(f_dbl(f_dbl(C)));
    r.Y = f_sub(f_mul(E, f_sub(D, r.X)), C8);
    r.Z = f_dbl(f_mul(p.Y, p.Z));
    return r;
}
// madd-2007-bl: Jacobian + affine (qx, qy) (affine point must not be infinity)
template <class F>
FQ_HD jacT<F> jac_madd(const jacT<F> &p, const F &qx, const F &qy) {
    if (f_is_zero(p.Z)) {
        jacT<F> r;
        r.X = qx;
        r.Y = qy;
        r.Z = FT<F>::one();
        return r;
    }
    F Z1Z1 = f_sqr(p.Z);
    F U2 = f_mul(qx, Z1Z1);
    F S2 = f_mul(f_mul(qy, p.Z), Z1Z1);
    if (f_eq(U2, p.X)) {
        if (f_eq(S2, p.Y)) return jac_dbl(p);
        return jac_inf<F>();
    }
    F H = f_sub(U2, p.X);
    F HH = f_sqr(H);
    F I = f_dbl(f_dbl(HH));
    F J = f_mul(H, I);
    F rr = f_dbl(f_sub(S2, p.Y));
    F V = f_mul(p.X, I);
    jacT<F> r;
    r.X = f_sub(f_sub(f_sqr(rr), J), f_dbl(V));
    r.Y = f_sub(f_mul(rr, f_sub(V, r.X)), f_dbl(f_mul(p.Y, J)));
    r.Z = f_sub(f_sub(f_sqr(f_add(p.Z, H)), Z1Z1), HH);
    return r;
}
// madd-2007-bl on lazy values (G1 bucket sums).  Invariant of the accumulator between calls (units of q):
//   X in N(7), Y in N(5), Z in W(2.3);  the affine point is canonical.  Bounds of every step, with lz_mul -> 1 + a b / 169:
//   Z1Z1, U2, t, S2 < 2;  H = U2 - X + 7q < 8.02;  HH < 1.38;  I = 4 HH < 5.52;  J = H I -> < 1.27;  V = X I -> < 1.23;
//   r0 = S2 - Y + 5q < 6.02, rr = 2 r0 < 12.03 (W), rr^2 -> < 1.86;  X3 = rr^2 - J - 2V + 4q < 5.86  (J + 2V < 3.73);
//   V - X3 + 6q < 7.23, rr (V - X3) -> < 1.52;  Y J -> < 1.04;  Y3 = .. - 2 Y J + 3q < 4.52;  Z3 = 2 (Z H) < 2.22 (W).
// H == 0 mod q (the point equals +-the accumulator: doubling or infinity) is detected on HH, which lz_mul returns in
// N(2) with exact limbs, and handled by the canonical code on canonicalised inputs.
__device__ __forceinline__ jacT<fq> jac_madd_lazy(const jacT<fq> &p, const fq &qx, const fq &qy) {
    if (fq_is_zero(p.Z)) {            // infinity is always the exact zero
        jacT<fq> r;
        r.X = qx;
        r.Y = qy;
        r.Z = fq_one();
        return r;
    }
    const fq Z1Z1 = lz_mul(p.Z, p.Z);
    const fq U2 = lz_mul(qx, Z1Z1);
    const fq S2 = lz_mul(lz_mul(qy, p.Z), Z1Z1);
    const fq H = lz_sub<7>(U2, p.X);
    const fq HH = lz_mul(H, H);
    if (lz_is_zero_mod_q(HH)) {
        jacT<fq> c;
        c.X = lz_canon(p.X);
        c.Y = lz_canon(p.Y);
        c.Z = lz_canon(p.Z);
        return jac_madd(c, qx, qy);   // canonical: doubles or returns the exact infinity
    }
    const fq I = lz_quad(HH);
    const fq J = lz_mul(H, I);
    const fq rr = lz_dbl(lz_sub<5>(S2, p.Y));
    const fq V = lz_mul(p.X, I);
    jacT<fq> r;
    r.X = lz_sub2<4>(lz_mul(rr, rr), J, lz_dbl(V));
    r.Y = lz_sub<3>(lz_mul(rr, lz_sub<6>(V, r.X)), lz_dbl(lz_mul(p.Y, J)));
    r.Z = lz_dbl(lz_mul(p.Z, H));
    return r;
}
__device__ __forceinline__ jacT<fq> jac_canon(const jacT<fq> &p) {
    jacT<fq> r;
    r.X = lz_canon(p.X);
    r.Y = lz_canon(p.Y);
    r.Z = lz_canon(p.Z);
    return r;
}
__device__ __forceinline__ jacT<fq2> jac_canon(const jacT<fq2> &p) { return p; }   // G2 sums stay canonical throughout

// add-2007-bl: Jacobian + Jacobian
template <class F>
FQ_HD jacT<F> jac_add(const jacT<F> &p, const jacT<F> &q) {
    if (f_is_zero(p.Z)) return q;
    if (f_is_zero(q.Z)) return p;
    F Z1Z1 = f_sqr(p.Z), Z2Z2 = f_sqr(q.Z);
    F U1 = f_mul(p.X, Z2Z2), U2 = f_mul(q.X, Z1Z1);
    F S1 = f_mul(f_mul(p.Y, q.Z), Z2Z2), S2 = f_mul(f_mul(q.Y, p.Z), Z1Z1);
    if (f_eq(U1, U2)) {
        if (f_eq(S1, S2)) return jac_dbl(p);
        return jac_inf<F>();
    }
    F H = f_sub(U2, U1);
    F I = f_sqr(f_dbl(H));
    F J = f_mul(H, I);
    F rr = f_dbl(f_sub(S2, S1));
    F V = f_mul(U1, I);
    jacT<F> r;
    r.X = f_sub(f_sub(f_sqr(rr), J), f_dbl(V));
    r.Y = f_sub(f_mul(rr, f_sub(V, r.X)), f_dbl(f_mul(S1, J)));
    r.Z = f_mul(f_sub(f_sub(f_sqr(f_add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
    return r;
}
template <class F>
FQ_HD jacT<F> jac_mul_small(const jacT<F> &p, u32 k) {  // k * p by double-and-add (k < 2^32)
    jacT<F> acc = jac_inf<F>();
    int top = 31;
    while (top > 0 && !((k >> top) & 1)) top--;   // skip the leading zero bits (doublings of the point at infinity)
    for (int i = top; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k >> i) & 1) acc = jac_add(acc, p);
    }
    return acc;
}

// Signed window digits.  With K = sum_w 2^(cd w + cd - 1) the unsigned digits u_w of s + K give  s = sum_w (u_w - 2^(cd-1)) 2^(cd w):
// digits d_w in [-2^(cd-1), 2^(cd-1)), so a window has 2^(cd-1) buckets |d| = 1 .. 2^(cd-1) (bucket index |d| - 1) and a negative
// digit adds -P.  One more bit of window for the same bucket memory: 14 windows of 19 bits instead of 15 of 18.
// sc9 = s + K as nine 32-bit words (carry-propagated once per scalar); returns |d| (0: skip) and the sign.
__device__ __forceinline__ void add_bias9(const u32 *sc, const u32 *K, u32 *s9) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        c += (u64)(j < 8 ? sc[j] : 0u) + K[j];
        s9[j] = (u32)c;
        c >>= 32;
    }
}
__device__ __forceinline__ u32 digit_key(const u32 *s9, int w, int cd, u32 &neg) {
    const int bit = w * cd;
    const int limb = bit >> 5, off = bit & 31;
    u64 v = s9[limb];
    if (limb + 1 < 9) v |= (u64)s9[limb + 1] << 32;
    const int d = (int)((u32)(v >> off) & ((1u << cd) - 1)) - (1 << (cd - 1));
    neg = d < 0 ? 1u : 0u;
    return (u32)(d < 0 ? -d : d);
}

// ---- 1. sort of the point indices by window digit, two levels so that every global write lands next to its
// neighbours.  (The first version was one atomic counting sort: 4-byte writes to random addresses, 16x write
// amplification and a returning global atomic per (point, window) -- as slow as the bucket sums themselves.)
//   coarse digit = LOW `hi` bits of the window digit, fine digit = the `lo` bits above them (lo <= 10): skewed scalars
//   (many equal or tiny values, the short top window) then land in coarse bins that hold a single digit each, which
//   the fine stage handles with one LDS atomic per wave instead of 64 colliding ones
//   1a. msm_coarse_hist : LDS histogram per 4096-point tile -> counts[window][coarse]
//   1b. msm_scan        : exclusive scan per window -> coarse starts
//   1c. msm_coarse_part : per tile, LDS ranks + one global reservation per (window, coarse bin) -> (index, fine
//                         digit) pairs grouped by coarse bin (order inside a bin is arbitrary)
//   1d. msm_fine_sort   : one workgroup per (window, coarse bin): counting sort by fine digit with LDS counters,
//                         inside that bin's contiguous (L2-sized) region; also emits starts/counts per bucket
#define MSM_TILE 4096
#define MSM_LO_MAX 10
struct SortGeo {
    int c, hi, lo, nwin, wgroup;   // c = hi + lo: bucket-index bits of a window; wgroup: windows per pass of the tile kernels (LDS budget)
    int cd;                        // digit width = c + 1 (signed digits)
    u32 K[9];                      // the recoding bias  sum_w 2^(cd w + cd - 1)
};
__global__ void __launch_bounds__(256) msm_coarse_hist_kernel(const u32 *scalars, u64 n, SortGeo g, int w0, u32 *ccounts) {
    extern __shared__ u32 lh[];   // [wgroup][2^hi]
    const int nbin = 1 << g.hi, nw = min(g.wgroup, g.nwin - w0);
    for (int i = threadIdx.x; i < nw * nbin; i += 256) lh[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * MSM_TILE;
    for (int k = 0; k < MSM_TILE / 256; k++) {
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < n) {
            u32 sc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sc[j] = scalars[i * 8 + j];
            u32 s9[9], neg;
            add_bias9(sc, g.K, s9);
            for (int w = 0; w < nw; w++) {
                const u32 d = digit_key(s9, w0 + w, g.cd, neg);
                if (d) atomicAdd(&lh[w * nbin + ((d - 1) & (nbin - 1))], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nw * nbin; i += 256)
        if (lh[i]) atomicAdd(&ccounts[(u64)(w0 + i / nbin) * nbin + (i % nbin)], lh[i]);
}
// exclusive scan of the 2^bits counters of one window (one block per window)
__global__ void __launch_bounds__(1024) msm_scan_kernel(const u32 *counts, u32 *starts, u32 *cursor, int bits) {
    __shared__ u32 part[1024];
    const u64 base = (u64)blockIdx.x << bits;
    const u32 nb = 1u << bits;
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = min(threadIdx.x * per, nb), hi = min(lo + per, nb);
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
__global__ void __launch_bounds__(256) msm_coarse_part_kernel(const u32 *scalars, u64 n, SortGeo g, int w0, u32 *ccursor,
                                                             u32 *pidx, u32 *pfine) {
    extern __shared__ u32 lh[];   // [wgroup][2^hi] counters, then [wgroup][2^hi] global bases
    const int nbin = 1 << g.hi, nw = min(g.wgroup, g.nwin - w0);
    u32 *lbase = lh + g.wgroup * nbin;
    for (int i = threadIdx.x; i < nw * nbin; i += 256) lh[i] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * MSM_TILE;
    for (int k = 0; k < MSM_TILE / 256; k++) {   // phase A: how many of this tile go to each (window, coarse bin)
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < n) {
            u32 sc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sc[j] = scalars[i * 8 + j];
            u32 s9[9], neg;
            add_bias9(sc, g.K, s9);
            for (int w = 0; w < nw; w++) {
                const u32 d = digit_key(s9, w0 + w, g.cd, neg);
                if (d) atomicAdd(&lh[w * nbin + ((d - 1) & (nbin - 1))], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nw * nbin; i += 256) {   // one global reservation per (window, coarse bin) of the tile
        const u32 cnt = lh[i];
        lbase[i] = cnt ? atomicAdd(&ccursor[(u64)(w0 + i / nbin) * nbin + (i % nbin)], cnt) : 0u;
        lh[i] = 0;
    }
    __syncthreads();
    for (int k = 0; k < MSM_TILE / 256; k++) {   // phase B: rank inside the tile's share of the bin, write the pair
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < n) {
            u32 sc[8];
#pragma unroll
            for (int j = 0; j < 8; j++) sc[j] = scalars[i * 8 + j];
            u32 s9[9], neg;
            add_bias9(sc, g.K, s9);
            for (int w = 0; w < nw; w++) {
                const u32 d = digit_key(s9, w0 + w, g.cd, neg);
                if (d) {
                    const int b = w * nbin + ((d - 1) & (nbin - 1));
                    const u32 pos = lbase[b] + atomicAdd(&lh[b], 1u);
                    pidx[(u64)(w0 + w) * n + pos] = (u32)i | (neg << 31);   // the sign of the digit travels with the index
                    pfine[(u64)(w0 + w) * n + pos] = (d - 1) >> g.hi;
                }
            }
        }
    }
}
// ---- 1d. fine stage, slice-parallel: a coarse bin is cut into slices of MSM_FSLICE elements, one workgroup each, so
// that a bin holding millions of elements (skewed scalars) is sorted by many workgroups.
//   msm_slices      : slice list (coarse bin, slice number) per (window, coarse bin)
//   msm_fine_hist   : LDS histogram of a slice by fine digit -> counts[bucket] (global atomics, one per digit present)
//   msm_fine_scan   : per coarse bin: exclusive scan of its <= 1024 bucket counts -> starts[bucket], cursor[bucket]
//   msm_fine_scatter: per slice: reserve cursor[bucket] once per digit present, rank in LDS, write the indices
// A wave whose 64 elements carry one and the same digit (the skewed case) issues one LDS atomic, not 64 colliding ones.
#define MSM_FSLICE 8192
__global__ void __launch_bounds__(256) msm_slices_kernel(const u32 *ccounts, u32 ncoarse, u32 *slice_count, uint2 *slices) {
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ncoarse) return;
    const u32 cc = ccounts[i];
    const u32 nsl = (cc + MSM_FSLICE - 1) / MSM_FSLICE;
    if (!nsl) return;
    const u32 base = atomicAdd(slice_count, nsl);
    for (u32 s = 0; s < nsl; s++) slices[base + s] = make_uint2(i, s);
}
// LDS histogram of one slice by fine digit (cnt must be zero on entry, nf entries)
__device__ __forceinline__ void slice_hist(const u32 *pf, u32 lo, u32 hi, u32 *cnt) {
    const int lane = threadIdx.x & 63;
    for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
        const u32 f = pf[i];
        const u64 act = __ballot(1);
        if (__ballot(f == (u32)__builtin_amdgcn_readfirstlane((int)f)) == act) {   // one digit in the whole wave
            if (lane == __ffsll((long long)act) - 1) atomicAdd(&cnt[f], (u32)__popcll(act));
        } else {
            atomicAdd(&cnt[f], 1u);
        }
    }
}
__global__ void __launch_bounds__(256) msm_fine_hist_kernel(const u32 *pfine, u64 n, SortGeo g, const uint2 *slices,
                                                           const u32 *cstarts, const u32 *ccounts, u32 *counts) {
    __shared__ u32 cnt[1 << MSM_LO_MAX];
    const int nbin = 1 << g.hi, nf = 1 << g.lo;
    const uint2 sl = slices[blockIdx.x];
    const u64 w = sl.x / nbin, bin = sl.x % nbin;
    const u32 cs = cstarts[sl.x], cc = ccounts[sl.x];
    const u32 lo = sl.y * MSM_FSLICE, hi = min(lo + (u32)MSM_FSLICE, cc);
    for (int i = threadIdx.x; i < nf; i += 256) cnt[i] = 0;
    __syncthreads();
    slice_hist(pfine + w * n + cs, lo, hi, cnt);
    __syncthreads();
    for (int f = threadIdx.x; f < nf; f += 256)
        if (cnt[f]) atomicAdd(&counts[(w << g.c) + ((u64)f << g.hi) + bin], cnt[f]);
}
__global__ void __launch_bounds__(1024) msm_fine_scan_kernel(SortGeo g, const u32 *cstarts, const u32 *counts, u32 *starts,
                                                            u32 *cursor) {
    __shared__ u32 part[1024];
    const int nbin = 1 << g.hi, nf = 1 << g.lo;
    const u64 w = blockIdx.x / nbin, bin = blockIdx.x % nbin;
    const u32 cs = cstarts[blockIdx.x];
    const u64 bucket = (w << g.c) + ((u64)threadIdx.x << g.hi) + bin;
    const u32 mine = (int)threadIdx.x < nf ? counts[bucket] : 0u;
    part[threadIdx.x] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const u32 v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    if ((int)threadIdx.x < nf) {
        const u32 at = cs + part[threadIdx.x] - mine;
        starts[bucket] = at;
        cursor[bucket] = at;
    }
}
__global__ void __launch_bounds__(256) msm_fine_scatter_kernel(const u32 *pidx, const u32 *pfine, u64 n, SortGeo g,
                                                              const uint2 *slices, const u32 *cstarts, const u32 *ccounts,
                                                              u32 *cursor, u32 *sorted) {
    __shared__ u32 cnt[1 << MSM_LO_MAX], lbase[1 << MSM_LO_MAX];
    const int nbin = 1 << g.hi, nf = 1 << g.lo;
    const uint2 sl = slices[blockIdx.x];
    const u64 w = sl.x / nbin, bin = sl.x % nbin;
    const u32 cs = cstarts[sl.x], cc = ccounts[sl.x];
    const u32 lo = sl.y * MSM_FSLICE, hi = min(lo + (u32)MSM_FSLICE, cc);
    const u32 *pi = pidx + w * n + cs, *pf = pfine + w * n + cs;
    for (int i = threadIdx.x; i < nf; i += 256) cnt[i] = 0;
    __syncthreads();
    slice_hist(pf, lo, hi, cnt);
    __syncthreads();
    for (int f = threadIdx.x; f < nf; f += 256) {   // one global reservation per digit present in the slice
        const u32 k = cnt[f];
        lbase[f] = k ? atomicAdd(&cursor[(w << g.c) + ((u64)f << g.hi) + bin], k) : 0u;
        cnt[f] = 0;
    }
    __syncthreads();
    u32 *out = sorted + w * n;
    const int lane = threadIdx.x & 63;
    for (u32 i = lo + threadIdx.x; i < hi; i += 256) {
        const u32 f = pf[i];
        const u64 act = __ballot(1);
        u32 rank;
        if (__ballot(f == (u32)__builtin_amdgcn_readfirstlane((int)f)) == act) {
            u32 base = 0;
            if (lane == __ffsll((long long)act) - 1) base = atomicAdd(&cnt[f], (u32)__popcll(act));
            base = (u32)__builtin_amdgcn_readfirstlane((int)base);
            rank = base + (u32)__popcll(act & ((1ULL << lane) - 1));
        } else {
            rank = atomicAdd(&cnt[f], 1u);
        }
        out[lbase[f] + rank] = pi[i];
    }
}

// a packed affine point = 2 * FT<F>::WORDS words = NV uint4 (G1: 4, G2: 8); (0, 0) encodes the point at infinity
template <class F>
__device__ __forceinline__ void unpack_point(const uint4 *q, F &x, F &y) {
    constexpr int NV = FT<F>::WORDS / 2;
    u32 w[FT<F>::WORDS * 2];
#pragma unroll
    for (int k = 0; k < NV; k++) {
        w[4 * k] = q[k].x;
        w[4 * k + 1] = q[k].y;
        w[4 * k + 2] = q[k].z;
        w[4 * k + 3] = q[k].w;
    }
    x = FT<F>::from_words(w);
    y = FT<F>::from_words(w + FT<F>::WORDS);
}
// ---- 2a. one-time conversion of the affine inputs to Montgomery form (16-byte vector accesses); stored packed
template <class F>
__global__ void __launch_bounds__(256) msm_to_mont_kernel(const uint4 *points, u64 n, uint4 *mont) {
    constexpr int NV = FT<F>::WORDS / 2;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint4 q[NV];
#pragma unroll
    for (int k = 0; k < NV; k++) q[k] = points[i * NV + k];
    F x, y;
    unpack_point<F>(q, x, y);
    u32 w[FT<F>::WORDS * 2];
    FT<F>::to_words(f_to_mont(x), w);   // (0,0) stays (0,0): still the infinity marker
    FT<F>::to_words(f_to_mont(y), w + FT<F>::WORDS);
#pragma unroll
    for (int k = 0; k < NV; k++) mont[i * NV + k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
}
// ---- 2b. bucket sums: lane = (window, bucket).  Software pipeline, two points per trip: while point k is added,
// point k+1 and the index of point k+2 are in flight.  All loads are unconditional (indices clamped to the bucket's
// last point) so that no loop-carried register needs a copy under an exec mask -- with a conditional prefetch the
// compiler parked a v_mov (and therefore an s_waitcnt) right behind every load and nothing was overlapped.
template <class F>
__device__ __forceinline__ jacT<F> madd_packed(const jacT<F> &acc, const uint4 *q, u32 neg) {
    F x, y;
    unpack_point<F>(q, x, y);
    if (f_is_zero(x) && f_is_zero(y)) return acc;
    // negative digit: add -P = (x, -y)
    if constexpr (std::is_same<F, fq>::value) {
        if (neg) y = lz_sub<1>(fq_zero(), y);  // q - y in (0, q]: one carry pass
        return jac_madd_lazy(acc, x, y);       // unreduced between the products; canonical at the store
    } else {
        if (neg) y = f_sub(FT<F>::zero(), y);
        return jac_madd(acc, x, y);
    }
}
// Buckets with more than MSM_HEAVY points are not summed by one lane: real scalars are not uniform (the top window of
// a 254-bit scalar has 2-12 significant bits, witnesses are full of small values), and one lane walking a million points
// would take seconds.  Such a bucket is cut into chunks of MSM_HCHUNK points, each summed by a whole workgroup
// (msm_heavy_kernel), and the chunk sums are added per bucket (msm_heavy_combine_kernel).
#define MSM_HEAVY 256
#define MSM_HCHUNK 16384
struct HeavyLists {
    u32 *counters;   // [0] = chunks appended, [1] = heavy buckets appended
    uint2 *chunks;   // (bucket id, chunk number)
    uint4 *heavy;    // (bucket id, first chunk slot, number of chunks, 0)
};
// Lanes of a wave sum buckets of EQUAL size: a wave runs as long as its largest bucket, and with ~4-64 points per bucket (Poisson) the largest
// of 64 is 1.3-1.6 times the mean -- that much of the bucket kernel was lanes waiting.  The (window, bucket) ids are therefore counting-sorted by
// their point count, largest first (bin MSM_HEAVY + 1: the heavy buckets, which only append to the heavy lists), and lane i takes order[i].
//   msm_order_hist : histogram of min(count, MSM_HEAVY + 1) over all buckets (LDS, then one global atomic per bin and block)
//   msm_order_scan : exclusive scan from the largest bin down (one block)
//   msm_order_fill : rank inside the block (LDS), one reservation per bin and block, order[pos] = id
#define MSM_OBINS (MSM_HEAVY + 2)
__global__ void __launch_bounds__(256) msm_order_hist_kernel(const u32 *counts, u64 nb, u32 *hist) {
    __shared__ u32 h[MSM_OBINS];
    for (int i = threadIdx.x; i < MSM_OBINS; i += 256) h[i] = 0;
    __syncthreads();
    for (u64 id = (u64)blockIdx.x * 256 + threadIdx.x; id < nb; id += (u64)gridDim.x * 256) {
        const u32 cnt = counts[id];
        atomicAdd(&h[cnt > MSM_HEAVY ? MSM_HEAVY + 1 : cnt], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MSM_OBINS; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}
__global__ void __launch_bounds__(64) msm_order_scan_kernel(const u32 *hist, u32 *cursor) {
    if (threadIdx.x) return;
    u32 run = 0;
    for (int b = MSM_OBINS - 1; b >= 0; b--) { cursor[b] = run; run += hist[b]; }
}
__global__ void __launch_bounds__(256) msm_order_fill_kernel(const u32 *counts, u64 nb, u32 *cursor, u32 *order) {
    __shared__ u32 h[MSM_OBINS], base[MSM_OBINS];
    const u64 id = (u64)blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < MSM_OBINS; i += 256) h[i] = 0;
    __syncthreads();
    u32 bin = 0, rank = 0;
    if (id < nb) {
        const u32 cnt = counts[id];
        bin = cnt > MSM_HEAVY ? MSM_HEAVY + 1 : cnt;
        rank = atomicAdd(&h[bin], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MSM_OBINS; i += 256)
        if (h[i]) base[i] = atomicAdd(&cursor[i], h[i]);
    __syncthreads();
    if (id < nb) order[base[bin] + rank] = (u32)id;
}
template <class F>
__global__ void __launch_bounds__(256) msm_bucket_kernel(const uint4 *mont, u64 n, int c, int nwin, const u32 *starts,
                                                        const u32 *counts, const u32 *sorted, const u32 *order, jacT<F> *buckets, HeavyLists hl) {
    constexpr int NV = FT<F>::WORDS / 2;
    const u64 gid = (u64)blockIdx.x * 256 + threadIdx.x;
    if (gid >= ((u64)nwin << c)) return;
    const u64 id = order[gid];
    const u64 w = id >> c;
    const u32 st = starts[id], cnt = counts[id];
    if (cnt > MSM_HEAVY) {
        const u32 nch = (cnt + MSM_HCHUNK - 1) / MSM_HCHUNK;
        const u32 slot = atomicAdd(&hl.counters[0], nch);
        for (u32 k = 0; k < nch; k++) hl.chunks[slot + k] = make_uint2((u32)id, k);
        hl.heavy[atomicAdd(&hl.counters[1], 1u)] = make_uint4((u32)id, slot, nch, 0u);
        return;
    }
    const u32 *idx = sorted + w * n + st;
    jacT<F> acc = jac_inf<F>();
    if (cnt) {
        const u32 last = cnt - 1;
        uint4 A[NV], B[NV];
        u32 va = idx[0];                     // bit 31: sign of the digit, bits 0..30: point index
        u64 pa = va & 0x7FFFFFFFu;
#pragma unroll
        for (int j = 0; j < NV; j++) A[j] = mont[pa * NV + j];
        u32 ia = idx[last < 1 ? last : 1];
        for (u32 k = 0; k < cnt; k += 2) {
            const u32 vb = ia;
            const u64 pb = vb & 0x7FFFFFFFu;
#pragma unroll
            for (int j = 0; j < NV; j++) B[j] = mont[pb * NV + j];
            const u32 ib = idx[k + 2 < last ? k + 2 : last];
            acc = madd_packed<F>(acc, A, va >> 31);
            va = ib;
            pa = va & 0x7FFFFFFFu;
#pragma unroll
            for (int j = 0; j < NV; j++) A[j] = mont[pa * NV + j];
            ia = idx[k + 3 < last ? k + 3 : last];
            if (k + 1 < cnt) acc = madd_packed<F>(acc, B, vb >> 31);
        }
    }
    buckets[id] = jac_canon(acc);
}
// one workgroup per chunk of a heavy bucket: lane t sums points t, t+256, ... of the chunk, LDS tree over the lanes
template <class F>
__global__ void __launch_bounds__(256) msm_heavy_kernel(const uint4 *mont, u64 n, int c, const u32 *starts, const u32 *counts,
                                                       const u32 *sorted, HeavyLists hl, jacT<F> *partial) {
    constexpr int NV = FT<F>::WORDS / 2;
    __shared__ jacT<F> sh[256];
    const uint2 ch = hl.chunks[blockIdx.x];
    const u64 id = ch.x, w = id >> c;
    const u32 cnt = counts[id];
    const u32 lo = ch.y * MSM_HCHUNK, hi = min(lo + (u32)MSM_HCHUNK, cnt);
    const u32 *idx = sorted + w * n + starts[id];
    jacT<F> acc = jac_inf<F>();
    for (u32 k = lo + threadIdx.x; k < hi; k += 256) {
        const u32 vi = idx[k];
        const u64 pi = vi & 0x7FFFFFFFu;
        uint4 q[NV];
#pragma unroll
        for (int j = 0; j < NV; j++) q[j] = mont[pi * NV + j];
        acc = madd_packed<F>(acc, q, vi >> 31);
    }
    sh[threadIdx.x] = jac_canon(acc);
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = jac_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = sh[0];
}
// one wave per heavy bucket: lane t adds chunk sums t, t+64, ..., LDS tree over the lanes
template <class F>
__global__ void __launch_bounds__(64) msm_heavy_combine_kernel(HeavyLists hl, u32 nheavy, const jacT<F> *partial, jacT<F> *buckets) {
    __shared__ jacT<F> sh[64];
    const uint4 h = hl.heavy[blockIdx.x];
    jacT<F> acc = jac_inf<F>();
    for (u32 k = threadIdx.x; k < h.z; k += 64) acc = jac_add(acc, partial[h.y + k]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 32; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = jac_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) buckets[h.x] = sh[0];
}
// ---- 3a. per segment of SEG buckets: sum_{b in seg} (b + 1) * B_b   (bucket index b holds the points of digit magnitude b + 1)
// (round 6: 16 instead of 64 -- the kernel is one dependent chain of additions per lane with fewer lanes than the chip has SIMD slots:
// four times the lanes at a third of the chain, 2.2 -> 0.6 ms per 2^24-point run)
#define MSM_SEG 16
template <class F>
__global__ void __launch_bounds__(64) msm_segment_kernel(const jacT<F> *buckets, int c, int nwin, jacT<F> *segs) {
    const u64 id = (u64)blockIdx.x * 64 + threadIdx.x;
    const u64 segs_per_win = (1ULL << c) / MSM_SEG;
    if (id >= (u64)nwin * segs_per_win) return;
    const u64 w = id / segs_per_win, sidx = id % segs_per_win;
    const u64 s = sidx * MSM_SEG;
    const jacT<F> *B = buckets + (w << c) + s;
    jacT<F> run = jac_inf<F>(), acc = jac_inf<F>();
    for (int k = MSM_SEG - 1; k >= 1; k--) {
        run = jac_add(run, B[k]);
        acc = jac_add(acc, run);
    }
    run = jac_add(run, B[0]);                       // total of the segment
    acc = jac_add(acc, jac_mul_small(run, (u32)s + 1));     // + (s + 1) * total
    segs[id] = acc;
}
// ---- 3b. tree sum of the segment results of one window (one block per window)
template <class F>
__global__ void __launch_bounds__(256) msm_window_kernel(const jacT<F> *segs, int nseg, jacT<F> *wins) {
    __shared__ jacT<F> sh[256];
    const jacT<F> *S = segs + (u64)blockIdx.x * nseg;
    jacT<F> acc = jac_inf<F>();
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
fq f_inv_host(const fq &a) { return fq_inv_host(a); }
fq2 f_inv_host(const fq2 &a) {  // (a0 - a1 u) / (a0^2 + a1^2)
    const fq d = fq_inv_host(fq_add(fq_sqr(a.c0), fq_sqr(a.c1)));
    return fq2_make(fq_mul(a.c0, d), fq_sub(fq_zero(), fq_mul(a.c1, d)));
}

// All scratch of a run comes from ONE ctx-owned arena that only grows: hipMalloc/hipFree of gigabytes per call cost
// up to 45 ms (more than the 2^22-point run itself) once the process holds other large allocations.
static int32_t msm_arena(zp_ctx *ctx, size_t bytes, char **out) {
    if (ctx->msm_arena_bytes < bytes) {
        if (ctx->msm_arena) {
            ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ZP_HIP(ctx, hipFree(ctx->msm_arena));
            ctx->msm_arena = nullptr;
            ctx->msm_arena_bytes = 0;
        }
        ZP_HIP(ctx, hipMalloc(&ctx->msm_arena, bytes));
        ctx->msm_arena_bytes = bytes;
    }
    *out = (char *)ctx->msm_arena;
    return ZP_OK;
}

// one Pippenger run over n points (n <= 2^24 from the entry points): result as a Jacobian point in *out
template <class F>
int32_t msm_chunk(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n, jacT<F> *out) {
    constexpr int NV = FT<F>::WORDS / 2;
    using J = jacT<F>;
    int c = 4;
    while (c < 16 && (1ULL << (c + 2)) <= n) c++;   // ~4 points per bucket up to c = 16
    while (c < 20 && (1ULL << (c + 8)) <= n) c++;   // wider windows only while buckets keep >= 128 points (signed digits: profiles/r2_msm_c_sweep.txt)
    if (ctx->tune_msm_c > 0) c = ctx->tune_msm_c;    // experiment knob
    if (c < 6) c = 6;                                // segments of MSM_SEG buckets (and the sort geometry) need c >= 6
    if (c > 22) c = 22;
    // c = bucket-index bits of a window; digits are signed and one bit wider (cd = c + 1).  s + K must stay below 2^(cd nwin)
    // for any 256-bit scalar (the BN254 group order has 254 bits): cd * nwin >= 258
    const int cd = c + 1;
    const int nwin = (258 + cd - 1) / cd;
    const u64 nb = (u64)nwin << c;
    u32 *d_counts = nullptr, *d_starts = nullptr, *d_sorted = nullptr;
    J *d_buckets = nullptr, *d_segs = nullptr, *d_wins = nullptr;
    uint4 *d_mont = nullptr;
    const u64 nseg = (1ULL << c) / MSM_SEG;
    SortGeo g;
    g.c = c;
    g.lo = c < MSM_LO_MAX ? c : MSM_LO_MAX;
    g.hi = c - g.lo;
    g.nwin = nwin;
    g.cd = cd;
    for (int j = 0; j < 9; j++) g.K[j] = 0;
    for (int w = 0; w < nwin; w++) {
        const int bit = cd * w + cd - 1;            // < 288
        g.K[bit >> 5] |= 1u << (bit & 31);
    }
    g.wgroup = nwin;
    while ((size_t)g.wgroup * ((size_t)2 << g.hi) * sizeof(u32) > 48 * 1024 && g.wgroup > 1) g.wgroup = (g.wgroup + 1) / 2;
    const u64 ncoarse = (u64)nwin << g.hi;
    ZP_HIP(ctx, hipSetDevice(ctx->device));
    // arena layout (256-byte aligned pieces): per-bucket counts/starts/cursor | coarse counts/starts/cursor | slice list |
    // sorted indices + coarse-partitioned (index, fine) pairs | Montgomery points | buckets/segments/windows | heavy lists | chunk sums
    const u64 max_slices = (u64)nwin * n / MSM_FSLICE + ncoarse + 1;
    const u64 max_heavy = (u64)nwin * n / MSM_HEAVY + 1, max_chunks = (u64)nwin * n / MSM_HCHUNK + max_heavy + 1;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t sz_counts = al((nb * 4 + 2 * MSM_OBINS + ncoarse * 3 + 8 + max_slices * 2) * 4), sz_sorted = al((u64)nwin * n * 4 * 3),
                 sz_mont = al((u64)n * NV * 16), sz_buckets = al((nb + nwin * nseg + nwin) * sizeof(J)),
                 sz_hl = al(16 + max_chunks * sizeof(uint2) + max_heavy * sizeof(uint4)), sz_partial = al(max_chunks * sizeof(J));
    char *arena = nullptr;
    ZP_TRY(msm_arena(ctx, sz_counts + sz_sorted + sz_mont + sz_buckets + sz_hl + sz_partial, &arena));
    d_counts = (u32 *)arena;
    d_sorted = (u32 *)(arena + sz_counts);
    d_mont = (uint4 *)(arena + sz_counts + sz_sorted);
    d_buckets = (J *)(arena + sz_counts + sz_sorted + sz_mont);
    u32 *d_hl = (u32 *)(arena + sz_counts + sz_sorted + sz_mont + sz_buckets);
    J *d_partial = (J *)(arena + sz_counts + sz_sorted + sz_mont + sz_buckets + sz_hl);
    d_starts = d_counts + nb;
    u32 *d_fcursor = d_starts + nb;
    u32 *d_order = d_fcursor + nb, *d_ohist = d_order + nb;     // bucket ids sorted by size | MSM_OBINS histogram bins, then as many cursors
    u32 *d_ccounts = d_ohist + 2 * MSM_OBINS, *d_cstarts = d_ccounts + ncoarse, *d_ccursor = d_cstarts + ncoarse;
    u32 *d_slice_count = d_ccursor + ncoarse;
    uint2 *d_slices = (uint2 *)(((uintptr_t)(d_slice_count + 4) + 7) & ~(uintptr_t)7);
    u32 *d_pidx = d_sorted + (u64)nwin * n, *d_pfine = d_pidx + (u64)nwin * n;
    d_segs = d_buckets + nb;
    d_wins = d_segs + nwin * nseg;
    ZP_HIP(ctx, hipMemsetAsync(d_ccounts, 0, (ncoarse * 3 + 4) * 4, ctx->stream));   // coarse counters + slice count
    const unsigned gb = (unsigned)((n + 255) / 256), gt = (unsigned)((n + MSM_TILE - 1) / MSM_TILE);
    hipLaunchKernelGGL(msm_to_mont_kernel<F>, dim3(gb), dim3(256), 0, ctx->stream, (const uint4 *)d_points, (u64)n, d_mont);
    const size_t lds1 = (size_t)g.wgroup * ((size_t)1 << g.hi) * sizeof(u32);
    for (int w0 = 0; w0 < nwin; w0 += g.wgroup)
        hipLaunchKernelGGL(msm_coarse_hist_kernel, dim3(gt), dim3(256), lds1, ctx->stream, (const u32 *)d_scalars, (u64)n, g, w0, d_ccounts);
    hipLaunchKernelGGL(msm_scan_kernel, dim3(nwin), dim3(1024), 0, ctx->stream, d_ccounts, d_cstarts, d_ccursor, g.hi);
    for (int w0 = 0; w0 < nwin; w0 += g.wgroup)
        hipLaunchKernelGGL(msm_coarse_part_kernel, dim3(gt), dim3(256), 2 * lds1, ctx->stream, (const u32 *)d_scalars, (u64)n, g, w0,
                           d_ccursor, d_pidx, d_pfine);
    // fine stage: slice list on the device, its length read back (one of the two host round trips of a run)
    hipLaunchKernelGGL(msm_slices_kernel, dim3((unsigned)((ncoarse + 255) / 256)), dim3(256), 0, ctx->stream, d_ccounts, (u32)ncoarse,
                       d_slice_count, d_slices);
    u32 nslices = 0;
    hipError_t he = hipMemcpyAsync(&nslices, d_slice_count, 4, hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    if (he == hipSuccess) he = hipMemsetAsync(d_counts, 0, nb * 4, ctx->stream);
    if (he == hipSuccess && nslices)
        hipLaunchKernelGGL(msm_fine_hist_kernel, dim3(nslices), dim3(256), 0, ctx->stream, d_pfine, (u64)n, g, d_slices, d_cstarts,
                           d_ccounts, d_counts);
    if (he == hipSuccess)
        hipLaunchKernelGGL(msm_fine_scan_kernel, dim3((unsigned)ncoarse), dim3(1024), 0, ctx->stream, g, d_cstarts, d_counts, d_starts,
                           d_fcursor);
    if (he == hipSuccess && nslices)
        hipLaunchKernelGGL(msm_fine_scatter_kernel, dim3(nslices), dim3(256), 0, ctx->stream, d_pidx, d_pfine, (u64)n, g, d_slices,
                           d_cstarts, d_ccounts, d_fcursor, d_sorted);
    if (he == hipSuccess) he = hipMemsetAsync(d_ohist, 0, 2 * MSM_OBINS * 4, ctx->stream);
    if (he == hipSuccess) {
        hipLaunchKernelGGL(msm_order_hist_kernel, dim3(256), dim3(256), 0, ctx->stream, (const u32 *)d_counts, nb, d_ohist);
        hipLaunchKernelGGL(msm_order_scan_kernel, dim3(1), dim3(64), 0, ctx->stream, (const u32 *)d_ohist, d_ohist + MSM_OBINS);
        hipLaunchKernelGGL(msm_order_fill_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream, (const u32 *)d_counts, nb, d_ohist + MSM_OBINS, d_order);
    }
    HeavyLists hl;
    hl.counters = d_hl;
    hl.heavy = (uint4 *)(d_hl + 4);
    hl.chunks = (uint2 *)(hl.heavy + max_heavy);
    ZP_HIP(ctx, hipMemsetAsync(d_hl, 0, 16, ctx->stream));
    hipLaunchKernelGGL(msm_bucket_kernel<F>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const uint4 *)d_mont, (u64)n, c, nwin, d_starts, d_counts, d_sorted, (const u32 *)d_order, d_buckets, hl);
    u32 hcnt[2] = {0, 0};
    if (he == hipSuccess) he = hipMemcpyAsync(hcnt, d_hl, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
    if (he == hipSuccess && hcnt[0]) {
        {
            hipLaunchKernelGGL(msm_heavy_kernel<F>, dim3(hcnt[0]), dim3(256), 0, ctx->stream, (const uint4 *)d_mont, (u64)n, c, d_starts,
                               d_counts, d_sorted, hl, d_partial);
            hipLaunchKernelGGL(msm_heavy_combine_kernel<F>, dim3(hcnt[1]), dim3(64), 0, ctx->stream, hl, hcnt[1], d_partial,
                               d_buckets);
        }
    }
    hipLaunchKernelGGL(msm_segment_kernel<F>, dim3((unsigned)((nwin * nseg + 63) / 64)), dim3(64), 0, ctx->stream, d_buckets, c, nwin, d_segs);
    hipLaunchKernelGGL(msm_window_kernel<F>, dim3(nwin), dim3(256), 0, ctx->stream, d_segs, (int)nseg, d_wins);
    hipError_t le = hipGetLastError();
    std::vector<J> wins(nwin);
    hipError_t ce = hipMemcpyAsync(wins.data(), d_wins, nwin * sizeof(J), hipMemcpyDeviceToHost, ctx->stream);
    hipError_t se = hipStreamSynchronize(ctx->stream);
    ZP_HIP(ctx, he);
    ZP_HIP(ctx, le);
    ZP_HIP(ctx, ce);
    ZP_HIP(ctx, se);
    // host: sum_w 2^(cd*w) * W_w  (Horner from the top window)
    J acc = jac_inf<F>();
    for (int w = nwin - 1; w >= 0; w--) {
        for (int k = 0; k < cd; k++) acc = jac_dbl(acc);
        acc = jac_add(acc, wins[w]);
    }
    *out = acc;
    return ZP_OK;
}

// Points are processed in runs of 2^24 (G1: 1 GiB of points): the bucket kernel reads points at random, and beyond
// that footprint its rate halves (2^26 in one run: 3.7 G additions/s against 8.1 G/s at 2^24); the partial sums are
// added on the host.
#define MSM_CHUNK_LOG 24
template <class F>
int32_t msm_run(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n, uint32_t *h_out) {
    constexpr int PW = FT<F>::WORDS * 2;   // words per affine point
    ZP_ARG(ctx, h_out != nullptr, "null output");
    ZP_ARG(ctx, n < (1ULL << 31), "too many points");
    memset(h_out, 0, PW * sizeof(uint32_t));
    if (n == 0) return ZP_OK;
    ZP_ARG(ctx, d_points && d_scalars, "null device pointer");
    jacT<F> acc = jac_inf<F>();
    const size_t chunk = (size_t)1 << (ctx->tune_msm_chunk_log > 0 ? ctx->tune_msm_chunk_log : MSM_CHUNK_LOG);
    for (size_t off = 0; off < n; off += chunk) {
        const size_t len = n - off < chunk ? n - off : chunk;
        jacT<F> part;
        ZP_TRY(msm_chunk<F>(ctx, d_points + off * PW, d_scalars + off * 8, len, &part));
        acc = jac_add(acc, part);
    }
    if (f_is_zero(acc.Z)) return ZP_OK;  // infinity: all-zero output
    F zi = f_inv_host(acc.Z);
    F zi2 = f_sqr(zi);
    F x = f_from_mont(f_mul(acc.X, zi2));
    F y = f_from_mont(f_mul(acc.Y, f_mul(zi2, zi)));
    FT<F>::to_words(x, h_out);
    FT<F>::to_words(y, h_out + FT<F>::WORDS);
    return ZP_OK;
}

}  // namespace

extern "C" int32_t zp_msm_bn254(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n,
                                uint32_t *h_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "msm_bn254");
    return msm_run<fq>(ctx, d_points, d_scalars, n, h_out);
}

extern "C" int32_t zp_msm_bn254_g2(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n,
                                   uint32_t *h_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "msm_bn254_g2");
    return msm_run<fq2>(ctx, d_points, d_scalars, n, h_out);
}

// ---- fixed-base multiplication: out_i = s_i * B for ONE base B -- the group elements of a Groth16 key ([u_j(tau)]_1, [v_j(tau)]_2, ...: millions of
// scalars times the generator; zp_r1cs_key_scalars makes the scalars).  An MSM sums; this does not.  Table T[w][d] = d 2^(8w) B (32 windows of
// 8 bits, 255 entries each, affine, Montgomery form, built on the host with one batched inversion), lane = scalar: 32 table additions
// (Jacobian += affine), result stored in Jacobian form; the host turns the results affine with batched inversions on threads (one field
// inversion per 1 024 points) and writes them in the layout zp_msm_bn254 / _g2 read.  Setup work: run once per key, not per proof.
namespace {

template <class F>
__global__ void __launch_bounds__(256) fixed_base_kernel(const uint4 *__restrict__ table, const u32 *__restrict__ scalars, u64 n, jacT<F> *__restrict__ out) {
    constexpr int NV = FT<F>::WORDS / 2;
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 sc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) sc[k] = scalars[i * 8 + k];
    jacT<F> acc = jac_inf<F>();
    for (int w = 0; w < 32; w++) {
        const u32 d = (sc[w >> 2] >> (8 * (w & 3))) & 255u;
        if (d == 0) continue;
        uint4 q[NV];
        const uint4 *src = table + ((size_t)w * 256 + d) * NV;
#pragma unroll
        for (int k = 0; k < NV; k++) q[k] = src[k];
        F x, y;
        unpack_point<F>(q, x, y);
        acc = jac_madd(acc, x, y);
    }
    out[i] = acc;
}

// Jacobian -> affine, standard form, packed words ((0, 0) for the point at infinity): Montgomery's trick over runs of `run` points
template <class F>
void batch_to_affine_host(const jacT<F> *pts, size_t n, u32 *out_words, int threads) {
    constexpr int W2 = 2 * FT<F>::WORDS;
    const size_t run = 1024, nrun = (n + run - 1) / run;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if (T > 32) T = 32;
    if ((size_t)T > nrun) T = (int)(nrun ? nrun : 1);
    auto work = [&](size_t r0, size_t r1) {
        std::vector<F> pre(run);
        for (size_t r = r0; r < r1; r++) {
            const size_t a = r * run, b = a + run < n ? a + run : n;
            F acc = FT<F>::one();
            for (size_t i = a; i < b; i++) {
                pre[i - a] = acc;
                if (!f_is_zero(pts[i].Z)) acc = f_mul(acc, pts[i].Z);
            }
            F inv = f_inv_host(acc);
            for (size_t i = b; i-- > a;) {
                u32 *o = out_words + i * W2;
                if (f_is_zero(pts[i].Z)) { memset(o, 0, W2 * 4); continue; }
                const F zi = f_mul(inv, pre[i - a]);          // 1 / Z_i
                inv = f_mul(inv, pts[i].Z);
                const F zi2 = f_sqr(zi);
                FT<F>::to_words(f_from_mont(f_mul(pts[i].X, zi2)), o);
                FT<F>::to_words(f_from_mont(f_mul(pts[i].Y, f_mul(zi2, zi))), o + FT<F>::WORDS);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++) {
        const size_t r0 = nrun * t / T, r1 = nrun * (t + 1) / T;
        if (r0 < r1) pool.emplace_back(work, r0, r1);
    }
    for (auto &th : pool) th.join();
}

template <class F>
int32_t fixed_base_run(zp_ctx *ctx, const uint32_t *h_base, const uint32_t *h_scalars, size_t n, uint32_t *h_points, int32_t threads) {
    constexpr int NV = FT<F>::WORDS / 2, W2 = 2 * FT<F>::WORDS;
    ZP_ARG(ctx, h_base && (n == 0 || (h_scalars && h_points)), "null pointer");
    if (n == 0) return ZP_OK;
    F bx = f_to_mont(FT<F>::from_words(h_base)), by = f_to_mont(FT<F>::from_words(h_base + FT<F>::WORDS));
    ZP_ARG(ctx, !(f_is_zero(bx) && f_is_zero(by)), "the base is the point at infinity");
    // the table in Jacobian form on the host: T[w][d] = T[w][d - 1] + B_w, B_(w+1) = 256 B_w
    std::vector<jacT<F>> tab((size_t)32 * 256);
    jacT<F> bw;
    bw.X = bx; bw.Y = by; bw.Z = FT<F>::one();
    for (int w = 0; w < 32; w++) {
        tab[(size_t)w * 256] = jac_inf<F>();
        for (int d = 1; d < 256; d++) tab[(size_t)w * 256 + d] = jac_add(tab[(size_t)w * 256 + d - 1], bw);
        bw = jac_add(tab[(size_t)w * 256 + 255], bw);
    }
    std::vector<u32> tw(tab.size() * W2), tm(tab.size() * W2);
    batch_to_affine_host<F>(tab.data(), tab.size(), tw.data(), threads);
    for (size_t e = 0; e < tab.size(); e++) {       // back to Montgomery form, the kernel's input (infinity stays (0, 0))
        FT<F>::to_words(f_to_mont(FT<F>::from_words(&tw[e * W2])), &tm[e * W2]);
        FT<F>::to_words(f_to_mont(FT<F>::from_words(&tw[e * W2 + FT<F>::WORDS])), &tm[e * W2 + FT<F>::WORDS]);
    }
    void *d_tab = nullptr, *d_sc = nullptr, *d_out = nullptr;
    int32_t rc = zp_dev_alloc(ctx, tm.size() * 4, &d_tab);
    if (rc == ZP_OK) rc = zp_dev_alloc(ctx, n * 32, &d_sc);
    if (rc == ZP_OK) rc = zp_dev_alloc(ctx, n * sizeof(jacT<F>), &d_out);
    if (rc == ZP_OK) rc = zp_h2d(ctx, d_tab, tm.data(), tm.size() * 4);
    if (rc == ZP_OK) rc = zp_h2d(ctx, d_sc, h_scalars, n * 32);
    std::vector<jacT<F>> res;
    if (rc == ZP_OK) {
        hipLaunchKernelGGL(fixed_base_kernel<F>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)d_tab, (const u32 *)d_sc, (u64)n,
                           (jacT<F> *)d_out);
        if (hipGetLastError() != hipSuccess) { ctx->err = "fixed_base_kernel launch failed"; rc = ZP_ERR_HIP; }
        (void)NV;
    }
    if (rc == ZP_OK) {
        res.resize(n);
        rc = zp_d2h(ctx, res.data(), d_out, n * sizeof(jacT<F>));
    }
    if (d_tab) (void)zp_dev_free(ctx, d_tab);
    if (d_sc) (void)zp_dev_free(ctx, d_sc);
    if (d_out) (void)zp_dev_free(ctx, d_out);
    if (rc != ZP_OK) return rc;
    batch_to_affine_host<F>(res.data(), n, h_points, threads);
    return ZP_OK;
}

}  // namespace

extern "C" int32_t zp_fixed_base_mul_bn254(zp_ctx *ctx, const uint32_t *h_base, const uint32_t *h_scalars, size_t n, uint32_t *h_points, int32_t threads) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "fixed_base_mul_bn254");
    try {
        return fixed_base_run<fq>(ctx, h_base, h_scalars, n, h_points, threads);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
}

extern "C" int32_t zp_fixed_base_mul_bn254_g2(zp_ctx *ctx, const uint32_t *h_base, const uint32_t *h_scalars, size_t n, uint32_t *h_points, int32_t threads) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "fixed_base_mul_bn254_g2");
    try {
        return fixed_base_run<fq2>(ctx, h_base, h_scalars, n, h_points, threads);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
}

// ---- synthetic MSM inputs: n DISTINCT points P_i = (start + i) * G of BN254 G1 (host code, this file's own field
// arithmetic).  Like zp_synth_trace it stands in for data the offline build cannot obtain (a real proving key); distinct
// points make an MSM benchmark read 64 B per point from HBM instead of hitting a small table in cache, and the known
// discrete logs give a checkable answer: sum_i s_i P_i = (sum_i s_i (start + i) mod r) * G.
// Affine chord additions P_base + j*G against a table of 1024 multiples, one batched inversion per block.
namespace {
struct affp { fq x, y; };
affp aff_add_host(const affp &p, const affp &q) {   // p != -q
    fq num, den;
    if (fq_eq(p.x, q.x)) {
        const fq t = fq_sqr(p.x);
        num = fq_add(fq_add(t, t), t);
        den = fq_add(p.y, p.y);
    } else {
        num = fq_sub(q.y, p.y);
        den = fq_sub(q.x, p.x);
    }
    const fq lam = fq_mul(num, fq_inv_host(den));
    affp r;
    r.x = fq_sub(fq_sub(fq_sqr(lam), p.x), q.x);
    r.y = fq_sub(fq_mul(lam, fq_sub(p.x, r.x)), p.y);
    return r;
}
affp aff_mul_g_host(const affp &g, u64 k) {   // k >= 1
    affp acc = g, base = g;
    bool started = false;
    for (int i = 0; i < 64 && (k >> i); i++) {
        if ((k >> i) & 1) {
            acc = started ? aff_add_host(acc, base) : base;
            started = true;
        }
        base = aff_add_host(base, base);
    }
    return acc;
}
void aff_store(uint32_t *out, const affp &p) {
    fq_to_words(fq_from_mont(p.x), out);
    fq_to_words(fq_from_mont(p.y), out + 8);
}
}  // namespace

#include <thread>
extern "C" int32_t zp_synth_g1_points(uint64_t start, size_t n, uint32_t *h_points, int32_t threads) {
    constexpr int BLK = 1024;
    if (!h_points || start <= (uint64_t)BLK || start + n < start) return ZP_ERR_ARG;
    if (n == 0) return ZP_OK;
    affp g;
    {
        fq one = fq_zero(), two = fq_zero();
        one.l[0] = 1;
        two.l[0] = 2;
        g.x = fq_to_mont(one);
        g.y = fq_to_mont(two);
    }
    std::vector<affp> tab(BLK + 1);   // tab[j] = j*G
    tab[1] = g;
    for (int j = 2; j <= BLK; j++) tab[j] = aff_add_host(tab[j - 1], g);
    const size_t nblk = (n + BLK - 1) / BLK;
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    if ((size_t)T > nblk) T = (int)nblk;
    auto work = [&](size_t b0, size_t b1) {
        std::vector<fq> den(BLK), pre(BLK);
        affp base = aff_mul_g_host(g, start + b0 * BLK);
        for (size_t b = b0; b < b1; b++) {
            const size_t cnt = (b + 1) * BLK <= n ? BLK : n - b * BLK;
            uint32_t *o = h_points + b * BLK * 16;
            aff_store(o, base);
            fq acc = fq_one();
            for (int j = 1; j <= BLK; j++) {
                den[j - 1] = fq_sub(tab[j].x, base.x);
                pre[j - 1] = acc;
                acc = fq_mul(acc, den[j - 1]);
            }
            fq inv = fq_inv_host(acc);
            affp next = base;
            for (int j = BLK; j >= 1; j--) {
                const fq dinv = fq_mul(inv, pre[j - 1]);
                inv = fq_mul(inv, den[j - 1]);
                if ((size_t)j >= cnt && j != BLK) continue;
                const fq lam = fq_mul(fq_sub(tab[j].y, base.y), dinv);
                affp r;
                r.x = fq_sub(fq_sub(fq_sqr(lam), base.x), tab[j].x);
                r.y = fq_sub(fq_mul(lam, fq_sub(base.x, r.x)), base.y);
                if (j == BLK) next = r;
                if ((size_t)j < cnt) aff_store(o + (size_t)j * 16, r);
            }
            base = next;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < T; t++) {
        const size_t b0 = nblk * t / T, b1 = nblk * (t + 1) / T;
        if (b0 < b1) pool.emplace_back(work, b0, b1);
    }
    for (auto &th : pool) th.join();
    return ZP_OK;
}
