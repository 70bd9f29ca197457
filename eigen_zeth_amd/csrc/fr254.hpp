// F_r of BN254 (the scalar field, r = 21888242871839275222246405745257275088548364400416034343698204186575808495617) for
// gfx950 device code and host setup: Montgomery form, R = 2^261, nine 29-bit limbs -- the layout csrc/msm.hip uses for
// the base field F_q (why 29 bits: a 64-bit column accumulator absorbs all 18 products of a column, one v_mad_u64_u32
// per product and no carry handling).  Values are always fully reduced and normalised between operations.
// Used by the BN128-hash mode (csrc/poseidon_bn254.hip).  No reference counterpart.
#pragma once
#include <stdint.h>
#include "gl.hpp"

#define FR_B 29
#define FR_MASK 0x1FFFFFFFu
#define FR_INV29 0x0FFFFFFFu   // -r^-1 mod 2^29
struct fr {
    u32 l[9];
};
GL_HD u32 fr_p(int i) {
    switch (i) {
        case 0: return 0x10000001u;
        case 1: return 0x1f0fac9fu;
        case 2: return 0x0e5c2450u;
        case 3: return 0x07d090f3u;
        case 4: return 0x1585d283u;
        case 5: return 0x02db40c0u;
        case 6: return 0x00a6e141u;
        case 7: return 0x0e5c2634u;
        default: return 0x0030644eu;
    }
}
GL_HD fr fr_zero() {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
GL_HD fr fr_one() {   // R mod r
    const u32 v[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = v[i];
    return r;
}
GL_HD fr fr_r2() {    // R^2 mod r
    const u32 v[9] = {0x05b69bd4u, 0x06170a5au, 0x020cddceu, 0x1db6310bu, 0x0e54d0ffu, 0x1cf855e3u, 0x1c15e103u, 0x07d09161u, 0x000a054au};
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = v[i];
    return r;
}
// t: limbs possibly unnormalised (each < 2^31), value < 2r  ->  normalised value mod r (branch-free)
GL_HD fr fr_norm_sub(const u32 *t) {
    u32 n[9], d[9];
    int cn = 0, cd = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int vn = (int)t[i] + cn;
        n[i] = (u32)vn & FR_MASK;
        cn = vn >> FR_B;
        const int vd = (int)t[i] - (int)fr_p(i) + cd;
        d[i] = (u32)vd & FR_MASK;
        cd = vd >> FR_B;
    }
    const u32 use_d = cd < 0 ? 0u : 0xFFFFFFFFu;
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (d[i] & use_d) | (n[i] & ~use_d);
    return r;
}
GL_HD fr fr_add(const fr &a, const fr &b) {
    u32 t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = a.l[i] + b.l[i];
    return fr_norm_sub(t);
}
GL_HD fr fr_sub(const fr &a, const fr &b) {
    u32 d[9], e[9];
    int cd = 0, ce = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int vd = (int)a.l[i] - (int)b.l[i] + cd;
        d[i] = (u32)vd & FR_MASK;
        cd = vd >> FR_B;
        const int ve = (int)a.l[i] - (int)b.l[i] + (int)fr_p(i) + ce;
        e[i] = (u32)ve & FR_MASK;
        ce = ve >> FR_B;
    }
    const u32 use_e = cd < 0 ? 0xFFFFFFFFu : 0u;
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = (e[i] & use_e) | (d[i] & ~use_e);
    return r;
}
// Montgomery product a*b/R mod r: product scanning, one 64-bit accumulator per column
GL_HD fr fr_mul(const fr &a, const fr &b) {
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (u64)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fr_p(k - i);
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (u64)m[k] * fr_p(0);
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (u64)a.l[i] * b.l[k - i];
            acc += (u64)m[i] * fr_p(k - i);
        }
        t[k - 9] = (u32)acc & FR_MASK;
        acc >>= FR_B;
    }
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}
// a*a/R mod r: the 36 cross products once, doubled through the left operand (2 a_i < 2^30: a column takes at most
// 4 doubled products + a square + 9 reduction terms, < 2^63) -- 45 limb products instead of 81
GL_HD fr fr_sqr(const fr &a) {
    u32 m[9], t[9], d[9];
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) d[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (u64)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fr_p(k - i);
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (u64)m[k] * fr_p(0);
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; i++) acc += (u64)d[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (u64)a.l[k / 2] * a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * fr_p(k - i);
        t[k - 9] = (u32)acc & FR_MASK;
        acc >>= FR_B;
    }
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}
// (a0 b0 + a1 b1 + a2 b2) / R mod r with ONE Montgomery reduction: the three products share the column accumulators
// (36 terms of < 2^58 per column stay below 2^64).  All limbs < 2^29; the sum is < 3 r^2 < R r, so the result is < 2 r.
GL_HD fr fr_mul3(const fr &a0, const fr &b0, const fr &a1, const fr &b1, const fr &a2, const fr &b2) {
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) {
            acc += (u64)a0.l[i] * b0.l[k - i];
            acc += (u64)a1.l[i] * b1.l[k - i];
            acc += (u64)a2.l[i] * b2.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fr_p(k - i);
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (u64)m[k] * fr_p(0);
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            acc += (u64)a0.l[i] * b0.l[k - i];
            acc += (u64)a1.l[i] * b1.l[k - i];
            acc += (u64)a2.l[i] * b2.l[k - i];
            acc += (u64)m[i] * fr_p(k - i);
        }
        t[k - 9] = (u32)acc & FR_MASK;
        acc >>= FR_B;
    }
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}
// sum_{n < N} x[n] * c[n] / R mod r with ONE Montgomery reduction, N <= 6: the N products share the column accumulators
// (at most 9 N + 9 terms of < 2^58 per column: 63 x 2^58 < 2^64 at N = 6).  c: N x 9 limbs, a wave-uniform table (scalar loads).
// The sum is < 6 r^2 and R > 169 r, so the reduced value is < 2 r and fr_norm_sub makes it canonical.
template <int N>
GL_HD fr fr_dotc(const fr *x, const u32 *c) {
    static_assert(N >= 1 && N <= 6, "at most six products per reduction");
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int n = 0; n < N; n++) {
#pragma unroll
            for (int i = 0; i <= k; i++) acc += (u64)x[n].l[i] * c[n * 9 + k - i];
        }
#pragma unroll
        for (int i = 0; i < k; i++) acc += (u64)m[i] * fr_p(k - i);
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (u64)m[k] * fr_p(0);
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int n = 0; n < N; n++) {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (u64)x[n].l[i] * c[n * 9 + k - i];
        }
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (u64)m[i] * fr_p(k - i);
        t[k - 9] = (u32)acc & FR_MASK;
        acc >>= FR_B;
    }
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}
GL_HD fr fr_to_mont(const fr &a) { return fr_mul(a, fr_r2()); }
GL_HD fr fr_from_mont(const fr &a) {
    fr one = fr_zero();
    one.l[0] = 1;
    return fr_mul(a, one);
}
// 4 x 64-bit words (little endian, value < r) <-> 9 x 29-bit limbs
GL_HD fr fr_from_u64(const u64 *w) {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = FR_B * i, k = bit >> 6, off = bit & 63;
        u64 v = w[k] >> off;
        if (off > 64 - FR_B && k + 1 < 4) v |= w[k + 1] << (64 - off);
        r.l[i] = (u32)v & FR_MASK;
    }
    return r;
}
GL_HD void fr_to_u64(const fr &a, u64 *w) {
#pragma unroll
    for (int k = 0; k < 4; k++) w[k] = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = FR_B * i, k = bit >> 6, off = bit & 63;
        w[k] |= (u64)a.l[i] << off;
        if (off > 64 - FR_B && k + 1 < 4) w[k + 1] |= (u64)a.l[i] >> (64 - off);
    }
}
GL_HD bool fr_is_canonical_u64(const u64 *w) {   // value < r
    const u64 R4[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    for (int k = 3; k >= 0; k--) {
        if (w[k] != R4[k]) return w[k] < R4[k];
    }
    return false;
}
