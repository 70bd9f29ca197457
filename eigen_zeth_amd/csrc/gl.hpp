// Goldilocks field  p = 2^64 - 2^32 + 1  for gfx950 device code and host-side plan setup.
//
// The reference (eigen-zeth) has no field arithmetic; this follows the public
// definition only.  2^64 == 2^32-1 (EPS) and 2^96 == -1 (mod p) give a multiply-free reduction
// of the 128-bit product.  All values passed between functions are canonical (< p).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD inline
#endif

typedef unsigned long long u64;
typedef unsigned int u32;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL

GL_HD u64 gl_canon(u64 s) {  // s in [0,2^64) -> s mod p   (s >= p  <=>  s + EPS wraps)
    u64 t = s + GL_EPS;
    return t < s ? t : s;
}
GL_HD u64 gl_add(u64 a, u64 b) {  // canonical in, canonical out (6 VALU on gfx950)
    u64 s = a + b, u = s + GL_EPS;    // u = s - p (mod 2^64)
    return ((s < a) | (u < s)) ? u : s;
}
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b, w = d + GL_P;
    return (a < b) ? w : d;
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

// lo + hl*2^64 + hh*2^96  ==  lo + hl*EPS - hh   (any lo, hl, hh) -> canonical.
// One v_mad_u64_u32 forms lo + hl*EPS; the carry (+2^64) and the borrow of "- hh" (-2^64) are
// folded with a single 3-way correction (k = carry - borrow in {-1,0,1}).
GL_HD u64 gl_reduce96(u64 lo, u32 hl, u32 hh) {
    u64 r = (u64)hl * 0xFFFFFFFFu + lo;
    bool c2 = r < lo;
    u64 r2 = r - hh;
    bool bb = r < (u64)hh;
    u64 u = r2 + GL_EPS;
    bool c3 = u < r2;
    u64 w = r2 + GL_P;
    bool cw = bb & !c2;
    bool cu = (c2 & !bb) | ((bb == c2) & c3);
    u64 t = cu ? u : r2;
    return cw ? w : t;
}
// x = c0 + c1*2^32 + c2*2^64 + c3*2^96
GL_HD u64 gl_reduce_limbs(u32 c0, u32 c1, u32 c2, u32 c3) { return gl_reduce96(((u64)c1 << 32) | c0, c2, c3); }

GL_HD u64 gl_mul(u64 a, u64 b) {  // any u64 inputs, canonical output
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return gl_reduce96(((u64)(u32)p10 << 32) | (u32)p00, (u32)p11, (u32)(p11 >> 32));
}

// x * 2^S mod p for a compile-time 0 < S < 96 (2 is a 192-th root of unity: 2^96 == -1, so every
// 64-th root of unity is a power of 8 and radix-16 butterflies need shifts only).  x canonical.
template <int S>
GL_HD u64 gl_mul_pow2(u64 x) {
    static_assert(S > 0 && S < 96, "shift out of range");
    constexpr int q = S / 32, r = S % 32;
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    u32 y0, y1, y2;
    if (r == 0) {
        y0 = x0; y1 = x1; y2 = 0;
    } else {
        y0 = x0 << r;
        y1 = (u32)(x >> (32 - r));
        y2 = x1 >> (32 - r);
    }
    if (q == 0) {  // (y1:y0) + y2*EPS
        const u64 lo = ((u64)y1 << 32) | y0;
        const u64 t = (u64)y2 * 0xFFFFFFFFu + lo;
        const u64 u = t + GL_EPS;
        return ((t < lo) | (u < t)) ? u : t;
    } else if (q == 1) {  // (y0:0) + y1*EPS - y2
        return gl_reduce96((u64)y0 << 32, y1, y2);
    } else {  // y0*EPS - (y2:y1)      [2^128 == -2^32]
        const u64 t = (u64)y0 * 0xFFFFFFFFu;
        const u64 m = ((u64)y2 << 32) | y1;
        const u64 d = t - m, w = d + GL_P;
        return (t < m) ? w : d;
    }
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

// ---- "weak" forms: any u64 in, any u64 out, value preserved mod p (no final canonicalisation).
// Safe wherever the consumer is another multiplication or a limb-wise accumulation.
GL_HD u64 gl_reduce96_weak(u64 lo, u32 hl, u32 hh) {
    u64 r = (u64)hl * 0xFFFFFFFFu + lo;
    r += (r < lo) ? GL_EPS : 0;     // +2^64 == +EPS; cannot wrap again (r < 2^64 - 2^33 after a wrap)
    const u64 d = r - hh;
    return d - ((r < (u64)hh) ? GL_EPS : 0);  // -2^64 == -EPS; d >= 2^64 - 2^32 after a borrow
}
GL_HD u64 gl_mul_weak(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return gl_reduce96_weak(((u64)(u32)p10 << 32) | (u32)p00, (u32)p11, (u32)(p11 >> 32));
}
// weak + canonical -> weak
GL_HD u64 gl_add_weak(u64 a_any, u64 b_canon) {
    u64 s = a_any + b_canon;
    return s + ((s < a_any) ? GL_EPS : 0);
}

// ---- unreduced dot-product accumulator: sum of up to 2^31 full products a*b, one reduction at the end.
// Device form (round 5): the sum is kept as THREE sums of 32 x 32-bit partial products -- low x low, the two mixed ones together, high x high --
// each 64 bits + a 32-bit count of carries: a multiply-accumulate is four v_mad_u64_u32 whose carry-outs the add-with-carry behind them takes
// (inline assembly: hipcc spends a 64-bit compare, a select and an add on `count += t < acc`), 8 instructions where the 160-bit form below
// -- kept for host code -- takes 22; the kernels that sum columns against weights (out-of-domain evaluation, DEEP quotient, the random linear
// combination of the constraints) were bound by them.  Nine registers instead of five.
#if defined(__HIP_DEVICE_COMPILE__)
struct gl_acc {
    u64 a, b, c;        // sum a0 b0, sum (a0 b1 + a1 b0), sum a1 b1   (mod 2^64)
    u32 oa, ob, oc;     // ... and how often each wrapped
};
__device__ __forceinline__ gl_acc gl_acc_zero() { return gl_acc{0, 0, 0, 0u, 0u, 0u}; }
__device__ __forceinline__ void gl_acc_mac32(u64 &acc, u32 &ovf, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(ovf) : "v"(x), "v"(y) : "vcc");
}
__device__ __forceinline__ void gl_acc_mac(gl_acc &s, u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    gl_acc_mac32(s.a, s.oa, a0, b0);
    gl_acc_mac32(s.b, s.ob, a0, b1);
    gl_acc_mac32(s.b, s.ob, a1, b0);
    gl_acc_mac32(s.c, s.oc, a1, b1);
}
// a + 2^64 oa + 2^32 (b + 2^64 ob) + 2^64 (c + 2^64 oc) mod p, canonical
__device__ __forceinline__ u64 gl_acc_reduce(const gl_acc &s) {
    const u64 xa = gl_reduce96(s.a, s.oa, 0u), xb = gl_reduce96(s.b, s.ob, 0u), xc = gl_reduce96(s.c, s.oc, 0u);
    return gl_add(gl_add(xa, gl_mul(xb, 1ULL << 32)), gl_mul(xc, GL_EPS));
}
#else
struct gl_acc {
    u64 lo, hi;
    u32 top;
};
GL_HD gl_acc gl_acc_zero() {
    gl_acc s;
    s.lo = 0;
    s.hi = 0;
    s.top = 0;
    return s;
}
GL_HD void gl_acc_mac(gl_acc &s, u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 p00 = (u64)a0 * b0;
    const u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    const u64 p10 = (u64)a1 * b0 + (u32)p01;
    const u64 phi = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    const u64 plo = ((u64)(u32)p10 << 32) | (u32)p00;
    s.lo += plo;
    const u64 c = s.lo < plo ? 1u : 0u;
    const u64 h1 = s.hi + phi;
    u32 c2 = h1 < phi ? 1u : 0u;
    const u64 h2 = h1 + c;
    c2 += h2 < c ? 1u : 0u;
    s.hi = h2;
    s.top += c2;
}
// lo + hi*2^64 + top*2^128 with 2^128 == -2^32  -> canonical
GL_HD u64 gl_acc_reduce(const gl_acc &s) {
    const u64 r = gl_reduce96(s.lo, (u32)s.hi, (u32)(s.hi >> 32));
    return gl_sub(r, (u64)s.top << 32);
}
#endif
GL_HD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
// a^(p-2), p-2 = 0xFFFFFFFE_FFFFFFFF = (2*(2^31-1))*2^32 + (2^32-1): addition chain on x^(2^k-1),
// 75 squarings + 10 multiplications (plain square-and-multiply needs 63 + 63)
GL_HD u64 gl_sqr_n(u64 x, int n) {
    for (int i = 0; i < n; i++) x = gl_mul(x, x);
    return x;
}
GL_HD u64 gl_inv(u64 a) {
    const u64 t2 = gl_mul(gl_sqr_n(a, 1), a);        // a^(2^2-1)
    const u64 t3 = gl_mul(gl_sqr_n(t2, 1), a);       // 2^3-1
    const u64 t4 = gl_mul(gl_sqr_n(t2, 2), t2);      // 2^4-1
    const u64 t7 = gl_mul(gl_sqr_n(t4, 3), t3);      // 2^7-1
    const u64 t8 = gl_mul(gl_sqr_n(t4, 4), t4);      // 2^8-1
    const u64 t15 = gl_mul(gl_sqr_n(t8, 7), t7);     // 2^15-1
    const u64 t16 = gl_mul(gl_sqr_n(t8, 8), t8);     // 2^16-1
    const u64 t31 = gl_mul(gl_sqr_n(t16, 15), t15);  // 2^31-1
    const u64 t32 = gl_mul(gl_sqr_n(t31, 1), a);     // 2^32-1
    const u64 hi = gl_sqr_n(t31, 1);                 // a^(2^32-2)
    return gl_mul(gl_sqr_n(hi, 32), t32);
}

// primitive 2^logn-th root derived from the configured 2^32-th root
GL_HD u64 gl_root(u64 root32, int logn) {
    u64 w = root32;
    for (int i = logn; i < 32; i++) w = gl_mul(w, w);
    return w;
}

// ---- cubic extension F_p[x]/(x^3 - x - 1)
struct e3 {
    u64 c[3];
};
GL_HD e3 e3_make(u64 a, u64 b, u64 c) {
    e3 r;
    r.c[0] = a; r.c[1] = b; r.c[2] = c;
    return r;
}
GL_HD e3 e3_add(e3 a, e3 b) { return e3_make(gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]), gl_add(a.c[2], b.c[2])); }
GL_HD e3 e3_sub(e3 a, e3 b) { return e3_make(gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]), gl_sub(a.c[2], b.c[2])); }
GL_HD e3 e3_scale(e3 a, u64 s) { return e3_make(gl_mul(a.c[0], s), gl_mul(a.c[1], s), gl_mul(a.c[2], s)); }
// inverse through the adjugate of the multiplication matrix of a (basis 1, t, t^2; t^3 = t + 1)
// a^-1 = adj / det: the two halves separately, so that callers can share one base-field inversion between several elements
GL_HD e3 e3_adj(e3 a, u64 *det) {
    const u64 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2];
    const u64 s02 = gl_add(a0, a2), s12 = gl_add(a1, a2);
    const u64 c00 = gl_sub(gl_mul(s02, s02), gl_mul(s12, a1));
    const u64 c01 = gl_sub(gl_mul(s12, a2), gl_mul(a1, s02));
    const u64 c02 = gl_sub(gl_mul(a1, a1), gl_mul(s02, a2));
    *det = gl_add(gl_add(gl_mul(a0, c00), gl_mul(a2, c01)), gl_mul(a1, c02));
    return e3_make(c00, c01, c02);
}
GL_HD e3 e3_inv(e3 a) {
    u64 det;
    const e3 adj = e3_adj(a, &det);
    return e3_scale(adj, gl_inv(det));
}
GL_HD e3 e3_mul(e3 a, e3 b) {
    u64 d0 = gl_mul(a.c[0], b.c[0]);
    u64 d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    u64 d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    u64 d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1]));
    u64 d4 = gl_mul(a.c[2], b.c[2]);
    // x^3 = x + 1, x^4 = x^2 + x
    return e3_make(gl_add(d0, d3), gl_add(gl_add(d1, d3), d4), gl_add(d2, d4));
}
