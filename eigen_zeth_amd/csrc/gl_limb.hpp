// Goldilocks values as four signed 24-bit-position limbs, for the register butterflies of the NTT passes (gfx950).
//
// Why.  p = 2^64 - 2^32 + 1 leaves no headroom in 64 bits, so a canonical modular add / sub is a carry chain, a compare and a
// correction: 10 instructions of the 4-cycle class per butterfly, and x * 2^(12 e) another 8-12.  But the
// plain 32-bit add / sub, logic and constant shifts issue in 2 cycles, and 2^96 == -1 (mod p): with B = 2^24,
//     x  =  l0 + l1 B + l2 B^2 + l3 B^3,   B^4 == -1,   l_i signed 32-bit with |l_i| < 2^28,
// is a redundant representation in which
//   * add / sub is four carry-free 32-bit adds (8 cycles instead of 20-24),
//   * x * 2^(24 j) is a renaming of limbs with sign changes that the next add / sub absorbs (free): every radix-16 twiddle with
//     an even exponent, i.e. all twiddles of the last three butterfly levels,
//   * x * 2^12 (the odd exponents: four of sixteen values, first level only) is  l_i' = ((l_i & 0xFFF) << 12) + (l_(i-1) >> 12),
//   * and the way back to a canonical 64-bit value costs nothing extra when a twiddle product follows: with the factor given as
//     W_i = w * B^i mod p (i < 4; for a table of powers of a root of unity these are FOUR ENTRIES OF THE SAME TABLE, 2^24 being an
//     8-th root of unity) the product is  x w = sum_i l_i W_i  =  L + H 2^32  with  L = sum l_i lo32(W_i),  H = sum l_i hi32(W_i):
//     eight v_mad_i64_i32 into two 64-bit accumulators whose start values (CL, CH, with CL + CH 2^32 == 0 mod p) make both sums
//     non-negative, then  (L1 + H0 : L0) + (H1 + carry) * EPS  -> canonical by the closing sequence of gl_mul: 16 instructions,
//     one fewer than the canonical product it replaces (gl_asm.hpp: 17), and the butterflies before it cost 40 % of theirs.
// The words of a factor are "balanced": W = wl + wh 2^32 with wl, wh in [-2^31, 2^31) (gl_l4_balance), so that both fit the signed
// operands of v_mad_i64_i32; |l_i| < 2^28 keeps |sum| < 2^61.
// Bit-identical to gl_mul / gl_mul_pow2 of gl.hpp (tests/test_gl_limb.py checks the host build of this header against them over
// random and edge values; the device build is checked by the NTT parity tests).  No reference counterpart.
#pragma once
#include "gl.hpp"

typedef long long i64;
typedef int i32;

struct gl_l4 {
    u32 l[4];   // two's complement; unsigned storage so that wrap-free adds are not signed-overflow UB
};
struct gl_w4 {   // a factor: balanced words of w, w B, w B^2, w B^3
    i32 lo[4], hi[4];
};

#define GL_L4_CL 0xBFFFFFFF40000001ULL   // == -(2^62 * 2^32) mod p, inside [2^61, 2^64 - 2^61]
#define GL_L4_CH 0x4000000000000000ULL

// canonical (or any) u64 -> limbs of 24 / 24 / 16 / 0 bits
GL_HD gl_l4 gl_l4_from(u64 x) {
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    gl_l4 r;
    r.l[0] = x0 & 0xFFFFFFu;
    r.l[1] = ((x0 >> 24) | (x1 << 8)) & 0xFFFFFFu;
    r.l[2] = x1 >> 16;
    r.l[3] = 0;
    return r;
}
GL_HD gl_l4 gl_l4_add(const gl_l4 &a, const gl_l4 &b) {
    gl_l4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
GL_HD gl_l4 gl_l4_sub(const gl_l4 &a, const gl_l4 &b) {
    gl_l4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) r.l[i] = a.l[i] - b.l[i];
    return r;
}
// x * B^J, J = 0..3: limb i moves to i + J, wrapping with a sign change (B^4 == -1)
template <int J>
GL_HD gl_l4 gl_l4_rot(const gl_l4 &a) {
    static_assert(J >= 0 && J < 4, "J in 0..3");
    gl_l4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int src = (i - J) & 3;
        r.l[i] = (i - J) < 0 ? 0u - a.l[src] : a.l[src];
    }
    return r;
}
// x * 2^12: the low 12 bits of a limb stay (shifted up), the rest moves one limb up
GL_HD gl_l4 gl_l4_shl12(const gl_l4 &a) {
    gl_l4 r;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32 up = (u32)((i32)a.l[(i + 3) & 3] >> 12);
        const u32 low = (a.l[i] & 0xFFFu) << 12;
        r.l[i] = i == 0 ? low - up : low + up;
    }
    return r;
}
// x * 2^(12 E), E = 0..7 (the radix-16 twiddles)
template <int E>
GL_HD gl_l4 gl_l4_mul_c16(const gl_l4 &a) {
    static_assert(E >= 0 && E < 8, "E in 0..7");
    if constexpr (E & 1) return gl_l4_rot<E / 2>(gl_l4_shl12(a));
    else return gl_l4_rot<E / 2>(a);
}

// canonical w -> balanced words: the representative of w in about (-p/2, p/2) as wl + wh 2^32, both words signed 32-bit
GL_HD void gl_l4_balance(u64 w, i32 &lo, i32 &hi) {
    u32 w0 = (u32)w, w1 = (u32)(w >> 32);
    if (w1 >= 0x7FFFFFFFu) {            // w - p  =  w + EPS - 2^64  (as a signed 64-bit value: w + EPS wrapped)
        const u64 t = w + GL_EPS;
        w0 = (u32)t;
        w1 = (u32)(t >> 32);
    }
    lo = (i32)w0;
    hi = (i32)(w1 + (w0 >> 31));        // the low word now counts as signed: carry its sign into the high word
}

// (l0:l1) = L, (h0:h1) = H, both non-negative 64-bit sums with H < 2^63:  L + H 2^32 mod p, canonical
GL_HD u64 gl_l4_fold(u64 L, u64 H) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 l0 = (u32)L, l1 = (u32)(L >> 32), h0 = (u32)H, h1 = (u32)(H >> 32);
    u32 r0, r1;
    u64 c, g;
    asm("v_add_co_u32 " GL_V1 ", %2, %5, %6\n\t"                      // V1 = L1 + H0              -> c
        "v_mov_b32 " GL_V0 ", %4\n\t"                                 // V0 = L0
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V2 ", %2, %7, 0, %2\n\t"                  // V2 = H1 + c  (< 2^32)
        "v_mad_u64_u32 " GL_P0 ", %2, " GL_V2 ", -1, " GL_P0 "\n\t"   // T = V2 * EPS + (V1:V0)    -> g
        "v_add_co_u32 " GL_V2 ", %3, " GL_V0 ", -1\n\t"               // u = T + EPS               -> h
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V3 ", %3, " GL_V1 ", 0, %3\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %2, %2, %3\n\t"                                     // g | h: take u
        "v_cndmask_b32 %0, " GL_V0 ", " GL_V2 ", %2\n\t"
        "v_cndmask_b32 %1, " GL_V1 ", " GL_V3 ", %2"
        : "=v"(r0), "=v"(r1), "=&s"(c), "=&s"(g)
        : "v"(l0), "v"(l1), "v"(h0), "v"(h1)
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "");
    return ((u64)r1 << 32) | r0;
#else
    const u32 l0 = (u32)L, l1 = (u32)(L >> 32), h0 = (u32)H, h1 = (u32)(H >> 32);
    const u64 v1 = (u64)l1 + h0;
    const u32 v2 = h1 + (u32)(v1 >> 32);
    const u64 lo = ((u64)(u32)v1 << 32) | l0;
    const u64 t = (u64)v2 * 0xFFFFFFFFu + lo;
    const u64 u = t + GL_EPS;
    return ((t < lo) | (u < t)) ? u : t;
#endif
}
// x * w mod p, canonical (x: |l_i| < 2^28)
GL_HD u64 gl_l4_mul(const gl_l4 &x, const gl_w4 &w) {
    i64 L = (i64)GL_L4_CL, H = (i64)GL_L4_CH;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        L += (i64)(i32)x.l[i] * (i64)w.lo[i];
        H += (i64)(i32)x.l[i] * (i64)w.hi[i];
    }
    return gl_l4_fold((u64)L, (u64)H);
}
// x mod p, canonical: the product with w = 1, whose words are (1, 0), (2^24, 0), (0, 2^16), (-2^8, 2^8)  [2^72 == 2^40 - 2^8]
GL_HD u64 gl_l4_canon(const gl_l4 &x) {
    const i64 L = (i64)GL_L4_CL + (i64)(i32)x.l[0] + (i64)(i32)x.l[1] * (i64)(1 << 24) - (i64)(i32)x.l[3] * (i64)(1 << 8);
    const i64 H = (i64)GL_L4_CH + (i64)(i32)x.l[2] * (i64)(1 << 16) + (i64)(i32)x.l[3] * (i64)(1 << 8);
    return gl_l4_fold((u64)L, (u64)H);
}
// the four balanced factors of a canonical w from w, w 2^24, w 2^48, w 2^72 (plan / table construction; not the hot loop)
GL_HD gl_w4 gl_l4_factor(u64 w) {
    gl_w4 r;
    gl_l4_balance(w, r.lo[0], r.hi[0]);
    gl_l4_balance(gl_mul_pow2<24>(w), r.lo[1], r.hi[1]);
    gl_l4_balance(gl_mul_pow2<48>(w), r.lo[2], r.hi[2]);
    gl_l4_balance(gl_mul_pow2<72>(w), r.lo[3], r.hi[3]);
    return r;
}
