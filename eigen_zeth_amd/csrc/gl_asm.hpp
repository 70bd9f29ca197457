// Hand-scheduled Goldilocks butterflies for gfx950 (device only).
//
// hipcc lowers gl_add / gl_sub (gl.hpp) to 64-bit adds + v_cmp_*_u64 + two selects: 6 VALU each, and it pads every
// VALU-writes-SGPR -> VALU-reads-SGPR pair (carry, compare mask) with s_nop because it keeps each chain contiguous.
// Here the same canonical results come from carry chains:
//     x - y : v_sub_co, v_subb_co            borrow br  => subtract EPS = 2^32 - 1 (i.e. add p mod 2^64):
//             lo += br (carry c2), hi -= br & ~c2                                   4 VALU + 1 SALU
//     x + y : v_add_co, v_addc_co  (carry c1);  u = s + EPS (carry c2  <=>  s >= p);
//             take u when c1 | c2                                                    6 VALU + 1 SALU
// and two butterflies are issued as four round-robin streams, so every SGPR written by a VALU instruction is read at
// least three instructions later (gfx950 needs two wait states there; inside an asm statement nobody pads them).
// Inputs and outputs are canonical (< p), bit-identical to gl_add / gl_sub.
#pragma once
#include "gl.hpp"

#if defined(__HIPCC__)
__device__ __forceinline__ void gl_bfly2(u64 &xa, u64 &ya, u64 &xb, u64 &yb) {
    u32 xa0 = (u32)xa, xa1 = (u32)(xa >> 32), ya0 = (u32)ya, ya1 = (u32)(ya >> 32);
    u32 xb0 = (u32)xb, xb1 = (u32)(xb >> 32), yb0 = (u32)yb, yb1 = (u32)(yb >> 32);
    u64 ca, fa, ea, ga, cb, fb, eb, gb;
    // temporaries: the fixed scratch registers gl_mul2 also uses (t = v116,v117 / v120,v121; u = v118,v119 / v122,v123),
    // so that the asm blocks of a kernel share one scratch window instead of each asking the allocator for its own
    asm("v_sub_co_u32 v116, %8, %0, %2\n\t"          //  1 S1a  t0 = x0 - y0            -> c
        "v_add_co_u32 %0, %10, %0, %2\n\t"           //  2 A1a  x0 = x0 + y0            -> e
        "v_sub_co_u32 v120, %12, %4, %6\n\t"         //  3 S1b
        "v_add_co_u32 %4, %14, %4, %6\n\t"           //  4 A1b
        "v_subb_co_u32 v117, %8, %1, %3, %8\n\t"     //  5 S2a  t1 = x1 - y1 - c        -> c = borrow
        "v_addc_co_u32 %1, %10, %1, %3, %10\n\t"     //  6 A2a  x1 = x1 + y1 + e        -> e = carry c1
        "v_subb_co_u32 v121, %12, %5, %7, %12\n\t"   //  7 S2b
        "v_addc_co_u32 %5, %14, %5, %7, %14\n\t"     //  8 A2b
        "v_addc_co_u32 %2, %9, v116, 0, %8\n\t"      //  9 S3a  y0 = t0 + borrow        -> f
        "v_add_co_u32 v118, %11, %0, -1\n\t"         // 10 A3a  u0 = x0 + 0xFFFFFFFF    -> g
        "v_addc_co_u32 %6, %13, v120, 0, %12\n\t"    // 11 S3b
        "v_add_co_u32 v122, %15, %4, -1\n\t"         // 12 A3b
        "s_andn2_b64 %8, %8, %9\n\t"                 // 13 S4a  c = borrow & ~f
        "v_addc_co_u32 v119, %11, %1, 0, %11\n\t"    // 14 A4a  u1 = x1 + g             -> g = carry c2
        "s_andn2_b64 %12, %12, %13\n\t"              // 15 S4b
        "v_addc_co_u32 v123, %15, %5, 0, %15\n\t"    // 16 A4b
        "v_subbrev_co_u32 %3, %9, 0, v117, %8\n\t"   // 17 S5a  y1 = t1 - c
        "s_or_b64 %10, %10, %11\n\t"                 // 18 A5a  e = c1 | c2
        "v_subbrev_co_u32 %7, %13, 0, v121, %12\n\t" // 19 S5b
        "s_or_b64 %14, %14, %15\n\t"                 // 20 A5b
        "v_cndmask_b32 %0, %0, v118, %10\n\t"        // 21 A6a
        "v_cndmask_b32 %1, %1, v119, %10\n\t"        // 22 A7a
        "v_cndmask_b32 %4, %4, v122, %14\n\t"        // 23 A6b
        "v_cndmask_b32 %5, %5, v123, %14"            // 24 A7b
        : "+v"(xa0), "+v"(xa1), "+v"(ya0), "+v"(ya1), "+v"(xb0), "+v"(xb1), "+v"(yb0), "+v"(yb1),   // 0..7
          "=&s"(ca), "=&s"(fa), "=&s"(ea), "=&s"(ga), "=&s"(cb), "=&s"(fb), "=&s"(eb), "=&s"(gb)     // 8..15
        :
        : "scc", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123");
    // (s_andn2 / s_or write SCC: without the clobber a loop branch scheduled across the block reads garbage)
    xa = ((u64)xa1 << 32) | xa0; ya = ((u64)ya1 << 32) | ya0;
    xb = ((u64)xb1 << 32) | xb0; yb = ((u64)yb1 << 32) | yb0;
}

// Two independent products a*b, c*d (any u64 inputs, canonical outputs), bit-identical to gl_mul.
// hipcc's gl_mul is 5 v_mad_u64_u32 + ~10 v_mov (zero-extending 32-bit halves into aligned 64-bit addend pairs) +
// 64-bit adds, five compares and four selects: ~100 issue cycles.  Here (70 cycles): three plain products and one
// accumulating one, the four limbs L0..L3 assembled with one carry chain, then  x = (L1:L0) - L3 + L2*EPS  as a
// borrow-corrected subtraction, ONE more mad whose carry-out and the carry of "+EPS" pick the canonical value.
// The mad results live in fixed scratch pairs v[116:127] so that their halves can be named (an inline-asm operand
// cannot name half of a 64-bit register pair); two products are interleaved and s_nop 0 pads the places where a carry
// would otherwise be read one instruction after it was written.
__device__ __forceinline__ void gl_mul2(u64 &a, u64 b, u64 &c, u64 d) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), d0 = (u32)d, d1 = (u32)(d >> 32);
    u32 ra0, ra1, rb0, rb1;
    u64 sd, cra, crb, ea, eb, fa, fb;
    asm("v_mad_u64_u32 v[116:117], %4, %11, %13, 0\n\t"          // PA = a0*b0
        "v_mad_u64_u32 v[122:123], %4, %15, %17, 0\n\t"          // PB
        "v_mad_u64_u32 v[118:119], %4, %11, %14, 0\n\t"          // RA = a0*b1
        "v_mad_u64_u32 v[124:125], %4, %15, %18, 0\n\t"
        "v_mad_u64_u32 v[120:121], %4, %12, %14, 0\n\t"          // HA = a1*b1
        "v_mad_u64_u32 v[126:127], %4, %16, %18, 0\n\t"
        "v_mad_u64_u32 v[118:119], %5, %12, %13, v[118:119]\n\t" // RA += a1*b0            -> cr
        "v_mad_u64_u32 v[124:125], %6, %16, %17, v[124:125]\n\t"
        "v_add_co_u32 v117, %7, v117, v118\n\t"                  // L1 = p1 + r0            -> c1
        "v_add_co_u32 v123, %8, v123, v124\n\t"
        "v_addc_co_u32 v121, %5, v121, 0, %5\n\t"                // L3 = h1 + cr
        "v_addc_co_u32 v127, %6, v127, 0, %6\n\t"
        "v_addc_co_u32 v120, %7, v120, v119, %7\n\t"             // L2 = h0 + r1 + c1       -> c2
        "v_addc_co_u32 v126, %8, v126, v125, %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 v121, %7, v121, 0, %7\n\t"                // L3 += c2
        "v_addc_co_u32 v127, %8, v127, 0, %8\n\t"
        "v_sub_co_u32 v116, %7, v116, v121\n\t"                  // t0 = L0 - L3            -> borrow
        "v_sub_co_u32 v122, %8, v122, v127\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 v117, %7, 0, v117, %7\n\t"             // t1 = L1 - borrow        -> borrow
        "v_subbrev_co_u32 v123, %8, 0, v123, %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 v116, %9, v116, 0, %7\n\t"                // borrowed 2^64 == EPS too much: t0 += 1 -> f
        "v_addc_co_u32 v122, %10, v122, 0, %8\n\t"
        "s_nop 0\n\t"
        "s_andn2_b64 %7, %7, %9\n\t"
        "s_andn2_b64 %8, %8, %10\n\t"
        "v_subbrev_co_u32 v117, %9, 0, v117, %7\n\t"             //                          t1 -= borrow & ~f
        "v_subbrev_co_u32 v123, %10, 0, v123, %8\n\t"
        "v_mad_u64_u32 v[116:117], %7, v120, -1, v[116:117]\n\t" // T = L2*EPS + t            -> g
        "v_mad_u64_u32 v[122:123], %8, v126, -1, v[122:123]\n\t"
        "v_add_co_u32 v118, %9, v116, -1\n\t"                    // u = T + EPS               -> h
        "v_add_co_u32 v124, %10, v122, -1\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 v119, %9, v117, 0, %9\n\t"
        "v_addc_co_u32 v125, %10, v123, 0, %10\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %7, %7, %9\n\t"                                // g | h: take u
        "s_or_b64 %8, %8, %10\n\t"
        "v_cndmask_b32 %0, v116, v118, %7\n\t"
        "v_cndmask_b32 %1, v117, v119, %7\n\t"
        "v_cndmask_b32 %2, v122, v124, %8\n\t"
        "v_cndmask_b32 %3, v123, v125, %8"
        : "=v"(ra0), "=v"(ra1), "=v"(rb0), "=v"(rb1),                                                  // 0..3
          "=&s"(sd), "=&s"(cra), "=&s"(crb), "=&s"(ea), "=&s"(eb), "=&s"(fa), "=&s"(fb)                // 4..10
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(d0), "v"(d1)                       // 11..18
        : "scc", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
    a = ((u64)ra1 << 32) | ra0;
    c = ((u64)rb1 << 32) | rb0;
}

// one product (odd counts): the same instruction sequence with s_nop 1 where the second product would have stood
__device__ __forceinline__ u64 gl_mul1(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 r0, r1;
    u64 sd, cr, e, f;
    asm("v_mad_u64_u32 v[116:117], %2, %6, %8, 0\n\t"
        "v_mad_u64_u32 v[118:119], %2, %6, %9, 0\n\t"
        "v_mad_u64_u32 v[120:121], %2, %7, %9, 0\n\t"
        "v_mad_u64_u32 v[118:119], %3, %7, %8, v[118:119]\n\t"
        "v_add_co_u32 v117, %4, v117, v118\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 v121, %3, v121, 0, %3\n\t"
        "v_addc_co_u32 v120, %4, v120, v119, %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 v121, %4, v121, 0, %4\n\t"
        "v_sub_co_u32 v116, %4, v116, v121\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 v117, %4, 0, v117, %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 v116, %5, v116, 0, %4\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %4, %4, %5\n\t"
        "v_subbrev_co_u32 v117, %5, 0, v117, %4\n\t"
        "v_mad_u64_u32 v[116:117], %4, v120, -1, v[116:117]\n\t"
        "v_add_co_u32 v118, %5, v116, -1\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 v119, %5, v117, 0, %5\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %4, %4, %5\n\t"
        "v_cndmask_b32 %0, v116, v118, %4\n\t"
        "v_cndmask_b32 %1, v117, v119, %4"
        : "=v"(r0), "=v"(r1), "=&s"(sd), "=&s"(cr), "=&s"(e), "=&s"(f)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "scc", "v116", "v117", "v118", "v119", "v120", "v121");
    return ((u64)r1 << 32) | r0;
}
#endif
