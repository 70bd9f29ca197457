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
GL_HD u64 gl_add(u64 a, u64 b) {  // canonical in, canonical out
    u64 s = a + b;
    if (s < a) s += GL_EPS;
    return gl_canon(s);
}
GL_HD u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    if (a < b) d -= GL_EPS;
    return d;
}
GL_HD u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }

// x = c0 + c1*2^32 + c2*2^64 + c3*2^96  ==  (c0 + c1*2^32) + c2*EPS - c3
GL_HD u64 gl_reduce_limbs(u32 c0, u32 c1, u32 c2, u32 c3) {
    u64 lo = ((u64)c1 << 32) | c0;
    u64 t0 = lo - c3;
    if (lo < (u64)c3) t0 -= GL_EPS;
    u64 t1 = ((u64)c2 << 32) - c2;  // c2 * EPS
    u64 r = t0 + t1;
    if (r < t1) r += GL_EPS;
    return gl_canon(r);
}

GL_HD u64 gl_mul(u64 a, u64 b) {
    u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p00 = (u64)a0 * b0;
    u64 p01 = (u64)a0 * b1 + (p00 >> 32);
    u64 p10 = (u64)a1 * b0 + (u32)p01;
    u64 p11 = (u64)a1 * b1 + (p01 >> 32) + (p10 >> 32);
    return gl_reduce_limbs((u32)p00, (u32)p10, (u32)p11, (u32)(p11 >> 32));
}
GL_HD u64 gl_sqr(u64 a) { return gl_mul(a, a); }

GL_HD u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
GL_HD u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

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
GL_HD e3 e3_mul(e3 a, e3 b) {
    u64 d0 = gl_mul(a.c[0], b.c[0]);
    u64 d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    u64 d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    u64 d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1]));
    u64 d4 = gl_mul(a.c[2], b.c[2]);
    // x^3 = x + 1, x^4 = x^2 + x
    return e3_make(gl_add(d0, d3), gl_add(gl_add(d1, d3), d4), gl_add(d2, d4));
}
