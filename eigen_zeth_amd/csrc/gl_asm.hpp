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

// Scratch window: twelve consecutive VGPRs named literally in the asm strings (an inline-asm operand cannot name half of
// a 64-bit register pair, and the mad results are needed by halves).  A translation unit picks the window with
// GL_ASM_SCRATCH_BASE before including this header: 116 (default: a kernel budgeted for 128 VGPRs, 4 waves per SIMD) or
// 52 (a kernel that fits 64 VGPRs, 8 waves per SIMD).  The base must be even (aligned pairs).
#ifndef GL_ASM_SCRATCH_BASE
#define GL_ASM_SCRATCH_BASE 116
#endif
#if GL_ASM_SCRATCH_BASE == 116
#define GL_V0 "v116"
#define GL_V1 "v117"
#define GL_V2 "v118"
#define GL_V3 "v119"
#define GL_V4 "v120"
#define GL_V5 "v121"
#define GL_V6 "v122"
#define GL_V7 "v123"
#define GL_V8 "v124"
#define GL_V9 "v125"
#define GL_V10 "v126"
#define GL_V11 "v127"
#define GL_P0 "v[116:117]"
#define GL_P2 "v[118:119]"
#define GL_P4 "v[120:121]"
#define GL_P6 "v[122:123]"
#define GL_P8 "v[124:125]"
#define GL_P10 "v[126:127]"
#elif GL_ASM_SCRATCH_BASE == 52
#define GL_V0 "v52"
#define GL_V1 "v53"
#define GL_V2 "v54"
#define GL_V3 "v55"
#define GL_V4 "v56"
#define GL_V5 "v57"
#define GL_V6 "v58"
#define GL_V7 "v59"
#define GL_V8 "v60"
#define GL_V9 "v61"
#define GL_V10 "v62"
#define GL_V11 "v63"
#define GL_P0 "v[52:53]"
#define GL_P2 "v[54:55]"
#define GL_P4 "v[56:57]"
#define GL_P6 "v[58:59]"
#define GL_P8 "v[60:61]"
#define GL_P10 "v[62:63]"
#else
#error "GL_ASM_SCRATCH_BASE must be 116 or 52"
#endif

#if defined(__HIPCC__)
__device__ __forceinline__ void gl_bfly2(u64 &xa, u64 &ya, u64 &xb, u64 &yb) {
    u32 xa0 = (u32)xa, xa1 = (u32)(xa >> 32), ya0 = (u32)ya, ya1 = (u32)(ya >> 32);
    u32 xb0 = (u32)xb, xb1 = (u32)(xb >> 32), yb0 = (u32)yb, yb1 = (u32)(yb >> 32);
    u64 ca, fa, ea, ga, cb, fb, eb, gb;
    // temporaries: the fixed scratch registers gl_mul2 also uses (t = v116,v117 / v120,v121; u = v118,v119 / v122,v123),
    // so that the asm blocks of a kernel share one scratch window instead of each asking the allocator for its own
    asm("v_sub_co_u32 " GL_V0 ", %8, %0, %2\n\t"          //  1 S1a  t0 = x0 - y0            -> c
        "v_add_co_u32 %0, %10, %0, %2\n\t"           //  2 A1a  x0 = x0 + y0            -> e
        "v_sub_co_u32 " GL_V4 ", %12, %4, %6\n\t"         //  3 S1b
        "v_add_co_u32 %4, %14, %4, %6\n\t"           //  4 A1b
        "v_subb_co_u32 " GL_V1 ", %8, %1, %3, %8\n\t"     //  5 S2a  t1 = x1 - y1 - c        -> c = borrow
        "v_addc_co_u32 %1, %10, %1, %3, %10\n\t"     //  6 A2a  x1 = x1 + y1 + e        -> e = carry c1
        "v_subb_co_u32 " GL_V5 ", %12, %5, %7, %12\n\t"   //  7 S2b
        "v_addc_co_u32 %5, %14, %5, %7, %14\n\t"     //  8 A2b
        "v_addc_co_u32 %2, %9, " GL_V0 ", 0, %8\n\t"      //  9 S3a  y0 = t0 + borrow        -> f
        "v_add_co_u32 " GL_V2 ", %11, %0, -1\n\t"         // 10 A3a  u0 = x0 + 0xFFFFFFFF    -> g
        "v_addc_co_u32 %6, %13, " GL_V4 ", 0, %12\n\t"    // 11 S3b
        "v_add_co_u32 " GL_V6 ", %15, %4, -1\n\t"         // 12 A3b
        "s_andn2_b64 %8, %8, %9\n\t"                 // 13 S4a  c = borrow & ~f
        "v_addc_co_u32 " GL_V3 ", %11, %1, 0, %11\n\t"    // 14 A4a  u1 = x1 + g             -> g = carry c2
        "s_andn2_b64 %12, %12, %13\n\t"              // 15 S4b
        "v_addc_co_u32 " GL_V7 ", %15, %5, 0, %15\n\t"    // 16 A4b
        "v_subbrev_co_u32 %3, %9, 0, " GL_V1 ", %8\n\t"   // 17 S5a  y1 = t1 - c
        "s_or_b64 %10, %10, %11\n\t"                 // 18 A5a  e = c1 | c2
        "v_subbrev_co_u32 %7, %13, 0, " GL_V5 ", %12\n\t" // 19 S5b
        "s_or_b64 %14, %14, %15\n\t"                 // 20 A5b
        "v_cndmask_b32 %0, %0, " GL_V2 ", %10\n\t"        // 21 A6a
        "v_cndmask_b32 %1, %1, " GL_V3 ", %10\n\t"        // 22 A7a
        "v_cndmask_b32 %4, %4, " GL_V6 ", %14\n\t"        // 23 A6b
        "v_cndmask_b32 %5, %5, " GL_V7 ", %14"            // 24 A7b
        : "+v"(xa0), "+v"(xa1), "+v"(ya0), "+v"(ya1), "+v"(xb0), "+v"(xb1), "+v"(yb0), "+v"(yb1),   // 0..7
          "=&s"(ca), "=&s"(fa), "=&s"(ea), "=&s"(ga), "=&s"(cb), "=&s"(fb), "=&s"(eb), "=&s"(gb)     // 8..15
        :
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "", "" GL_V6 "", "" GL_V7 "");
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
    asm("v_mad_u64_u32 " GL_P0 ", %4, %11, %13, 0\n\t"          // PA = a0*b0
        "v_mad_u64_u32 " GL_P6 ", %4, %15, %17, 0\n\t"          // PB
        "v_mad_u64_u32 " GL_P2 ", %4, %11, %14, 0\n\t"          // RA = a0*b1
        "v_mad_u64_u32 " GL_P8 ", %4, %15, %18, 0\n\t"
        "v_mad_u64_u32 " GL_P4 ", %4, %12, %14, 0\n\t"          // HA = a1*b1
        "v_mad_u64_u32 " GL_P10 ", %4, %16, %18, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %5, %12, %13, " GL_P2 "\n\t" // RA += a1*b0            -> cr
        "v_mad_u64_u32 " GL_P8 ", %6, %16, %17, " GL_P8 "\n\t"
        "v_add_co_u32 " GL_V1 ", %7, " GL_V1 ", " GL_V2 "\n\t"                  // L1 = p1 + r0            -> c1
        "v_add_co_u32 " GL_V7 ", %8, " GL_V7 ", " GL_V8 "\n\t"
        "v_addc_co_u32 " GL_V5 ", %5, " GL_V5 ", 0, %5\n\t"                // L3 = h1 + cr
        "v_addc_co_u32 " GL_V11 ", %6, " GL_V11 ", 0, %6\n\t"
        "v_addc_co_u32 " GL_V4 ", %7, " GL_V4 ", " GL_V3 ", %7\n\t"             // L2 = h0 + r1 + c1       -> c2
        "v_addc_co_u32 " GL_V10 ", %8, " GL_V10 ", " GL_V9 ", %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V5 ", %7, " GL_V5 ", 0, %7\n\t"                // L3 += c2
        "v_addc_co_u32 " GL_V11 ", %8, " GL_V11 ", 0, %8\n\t"
        "v_sub_co_u32 " GL_V0 ", %7, " GL_V0 ", " GL_V5 "\n\t"                  // t0 = L0 - L3            -> borrow
        "v_sub_co_u32 " GL_V6 ", %8, " GL_V6 ", " GL_V11 "\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %7, 0, " GL_V1 ", %7\n\t"             // t1 = L1 - borrow        -> borrow
        "v_subbrev_co_u32 " GL_V7 ", %8, 0, " GL_V7 ", %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V0 ", %9, " GL_V0 ", 0, %7\n\t"                // borrowed 2^64 == EPS too much: t0 += 1 -> f
        "v_addc_co_u32 " GL_V6 ", %10, " GL_V6 ", 0, %8\n\t"
        "s_nop 0\n\t"
        "s_andn2_b64 %7, %7, %9\n\t"
        "s_andn2_b64 %8, %8, %10\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %9, 0, " GL_V1 ", %7\n\t"             //                          t1 -= borrow & ~f
        "v_subbrev_co_u32 " GL_V7 ", %10, 0, " GL_V7 ", %8\n\t"
        "v_mad_u64_u32 " GL_P0 ", %7, " GL_V4 ", -1, " GL_P0 "\n\t" // T = L2*EPS + t            -> g
        "v_mad_u64_u32 " GL_P6 ", %8, " GL_V10 ", -1, " GL_P6 "\n\t"
        "v_add_co_u32 " GL_V2 ", %9, " GL_V0 ", -1\n\t"                    // u = T + EPS               -> h
        "v_add_co_u32 " GL_V8 ", %10, " GL_V6 ", -1\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V3 ", %9, " GL_V1 ", 0, %9\n\t"
        "v_addc_co_u32 " GL_V9 ", %10, " GL_V7 ", 0, %10\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %7, %7, %9\n\t"                                // g | h: take u
        "s_or_b64 %8, %8, %10\n\t"
        "v_cndmask_b32 %0, " GL_V0 ", " GL_V2 ", %7\n\t"
        "v_cndmask_b32 %1, " GL_V1 ", " GL_V3 ", %7\n\t"
        "v_cndmask_b32 %2, " GL_V6 ", " GL_V8 ", %8\n\t"
        "v_cndmask_b32 %3, " GL_V7 ", " GL_V9 ", %8"
        : "=v"(ra0), "=v"(ra1), "=v"(rb0), "=v"(rb1),                                                  // 0..3
          "=&s"(sd), "=&s"(cra), "=&s"(crb), "=&s"(ea), "=&s"(eb), "=&s"(fa), "=&s"(fb)                // 4..10
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(d0), "v"(d1)                       // 11..18
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "", "" GL_V6 "", "" GL_V7 "", "" GL_V8 "", "" GL_V9 "", "" GL_V10 "", "" GL_V11 "");
    a = ((u64)ra1 << 32) | ra0;
    c = ((u64)rb1 << 32) | rb0;
}

// one product (odd counts): the same instruction sequence with s_nop 1 where the second product would have stood
__device__ __forceinline__ u64 gl_mul1(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 r0, r1;
    u64 sd, cr, e, f;
    asm("v_mad_u64_u32 " GL_P0 ", %2, %6, %8, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %2, %6, %9, 0\n\t"
        "v_mad_u64_u32 " GL_P4 ", %2, %7, %9, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %3, %7, %8, " GL_P2 "\n\t"
        "v_add_co_u32 " GL_V1 ", %4, " GL_V1 ", " GL_V2 "\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V5 ", %3, " GL_V5 ", 0, %3\n\t"
        "v_addc_co_u32 " GL_V4 ", %4, " GL_V4 ", " GL_V3 ", %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V5 ", %4, " GL_V5 ", 0, %4\n\t"
        "v_sub_co_u32 " GL_V0 ", %4, " GL_V0 ", " GL_V5 "\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %4, 0, " GL_V1 ", %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V0 ", %5, " GL_V0 ", 0, %4\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %4, %4, %5\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %5, 0, " GL_V1 ", %4\n\t"
        "v_mad_u64_u32 " GL_P0 ", %4, " GL_V4 ", -1, " GL_P0 "\n\t"
        "v_add_co_u32 " GL_V2 ", %5, " GL_V0 ", -1\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V3 ", %5, " GL_V1 ", 0, %5\n\t"
        "s_nop 1\n\t"
        "s_or_b64 %4, %4, %5\n\t"
        "v_cndmask_b32 %0, " GL_V0 ", " GL_V2 ", %4\n\t"
        "v_cndmask_b32 %1, " GL_V1 ", " GL_V3 ", %4"
        : "=v"(r0), "=v"(r1), "=&s"(sd), "=&s"(cr), "=&s"(e), "=&s"(f)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "");
    return ((u64)r1 << 32) | r0;
}

// Weak forms (Poseidon: consumers take any u64): the same sequence without the final canonical choice -- the carry of the
// last mad is folded back as +EPS (T <= 2^64 - 2^33 after a wrap, so T + EPS cannot wrap again).  15 VALU instead of 17.
// The result is congruent to a*b mod p and < 2^64, NOT necessarily < p.
__device__ __forceinline__ void gl_mul2w(u64 &a, u64 b, u64 &c, u64 d) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u32 c0 = (u32)c, c1 = (u32)(c >> 32), d0 = (u32)d, d1 = (u32)(d >> 32);
    u32 ra0, ra1, rb0, rb1;
    u64 sd, cra, crb, ea, eb, fa, fb;
    asm("v_mad_u64_u32 " GL_P0 ", %4, %11, %13, 0\n\t"          // PA = a0*b0
        "v_mad_u64_u32 " GL_P6 ", %4, %15, %17, 0\n\t"          // PB
        "v_mad_u64_u32 " GL_P2 ", %4, %11, %14, 0\n\t"          // RA = a0*b1
        "v_mad_u64_u32 " GL_P8 ", %4, %15, %18, 0\n\t"
        "v_mad_u64_u32 " GL_P4 ", %4, %12, %14, 0\n\t"          // HA = a1*b1
        "v_mad_u64_u32 " GL_P10 ", %4, %16, %18, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %5, %12, %13, " GL_P2 "\n\t" // RA += a1*b0            -> cr
        "v_mad_u64_u32 " GL_P8 ", %6, %16, %17, " GL_P8 "\n\t"
        "v_add_co_u32 " GL_V1 ", %7, " GL_V1 ", " GL_V2 "\n\t"                  // L1 = p1 + r0            -> c1
        "v_add_co_u32 " GL_V7 ", %8, " GL_V7 ", " GL_V8 "\n\t"
        "v_addc_co_u32 " GL_V5 ", %5, " GL_V5 ", 0, %5\n\t"                // L3 = h1 + cr
        "v_addc_co_u32 " GL_V11 ", %6, " GL_V11 ", 0, %6\n\t"
        "v_addc_co_u32 " GL_V4 ", %7, " GL_V4 ", " GL_V3 ", %7\n\t"             // L2 = h0 + r1 + c1       -> c2
        "v_addc_co_u32 " GL_V10 ", %8, " GL_V10 ", " GL_V9 ", %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V5 ", %7, " GL_V5 ", 0, %7\n\t"                // L3 += c2
        "v_addc_co_u32 " GL_V11 ", %8, " GL_V11 ", 0, %8\n\t"
        "v_sub_co_u32 " GL_V0 ", %7, " GL_V0 ", " GL_V5 "\n\t"                  // t0 = L0 - L3            -> borrow
        "v_sub_co_u32 " GL_V6 ", %8, " GL_V6 ", " GL_V11 "\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %7, 0, " GL_V1 ", %7\n\t"             // t1 = L1 - borrow        -> borrow
        "v_subbrev_co_u32 " GL_V7 ", %8, 0, " GL_V7 ", %8\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V0 ", %9, " GL_V0 ", 0, %7\n\t"                // borrowed 2^64 == EPS too much: t0 += 1 -> f
        "v_addc_co_u32 " GL_V6 ", %10, " GL_V6 ", 0, %8\n\t"
        "s_nop 0\n\t"
        "s_andn2_b64 %7, %7, %9\n\t"
        "s_andn2_b64 %8, %8, %10\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %9, 0, " GL_V1 ", %7\n\t"             //                          t1 -= borrow & ~f
        "v_subbrev_co_u32 " GL_V7 ", %10, 0, " GL_V7 ", %8\n\t"
        "v_mad_u64_u32 " GL_P0 ", %7, " GL_V4 ", -1, " GL_P0 "\n\t" // T = L2*EPS + t            -> g
        "v_mad_u64_u32 " GL_P6 ", %8, " GL_V10 ", -1, " GL_P6 "\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %0, %9, 0, " GL_V0 ", %7\n\t"                 // 2^64 == EPS dropped by the mad: lo = T_lo - g -> h
        "v_subbrev_co_u32 %2, %10, 0, " GL_V6 ", %8\n\t"
        "s_nop 0\n\t"
        "s_andn2_b64 %7, %7, %9\n\t"
        "s_andn2_b64 %8, %8, %10\n\t"
        "v_addc_co_u32 %1, %9, " GL_V1 ", 0, %7\n\t"                    // hi = T_hi + (g & ~h)   (T + EPS cannot wrap again)
        "v_addc_co_u32 %3, %10, " GL_V7 ", 0, %8"
        : "=v"(ra0), "=v"(ra1), "=v"(rb0), "=v"(rb1),                                                  // 0..3
          "=&s"(sd), "=&s"(cra), "=&s"(crb), "=&s"(ea), "=&s"(eb), "=&s"(fa), "=&s"(fb)                // 4..10
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(c0), "v"(c1), "v"(d0), "v"(d1)                       // 11..18
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "", "" GL_V6 "", "" GL_V7 "", "" GL_V8 "", "" GL_V9 "", "" GL_V10 "", "" GL_V11 "");
    a = ((u64)ra1 << 32) | ra0;
    c = ((u64)rb1 << 32) | rb0;
}


__device__ __forceinline__ u64 gl_mul1w(u64 a, u64 b) {
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u32 r0, r1;
    u64 sd, cr, e, f;
    asm("v_mad_u64_u32 " GL_P0 ", %2, %6, %8, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %2, %6, %9, 0\n\t"
        "v_mad_u64_u32 " GL_P4 ", %2, %7, %9, 0\n\t"
        "v_mad_u64_u32 " GL_P2 ", %3, %7, %8, " GL_P2 "\n\t"
        "v_add_co_u32 " GL_V1 ", %4, " GL_V1 ", " GL_V2 "\n\t"
        "s_nop 0\n\t"
        "v_addc_co_u32 " GL_V5 ", %3, " GL_V5 ", 0, %3\n\t"
        "v_addc_co_u32 " GL_V4 ", %4, " GL_V4 ", " GL_V3 ", %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V5 ", %4, " GL_V5 ", 0, %4\n\t"
        "v_sub_co_u32 " GL_V0 ", %4, " GL_V0 ", " GL_V5 "\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %4, 0, " GL_V1 ", %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32 " GL_V0 ", %5, " GL_V0 ", 0, %4\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %4, %4, %5\n\t"
        "v_subbrev_co_u32 " GL_V1 ", %5, 0, " GL_V1 ", %4\n\t"
        "v_mad_u64_u32 " GL_P0 ", %4, " GL_V4 ", -1, " GL_P0 "\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %0, %5, 0, " GL_V0 ", %4\n\t"
        "s_nop 1\n\t"
        "s_andn2_b64 %4, %4, %5\n\t"
        "v_addc_co_u32 %1, %5, " GL_V1 ", 0, %4"
        : "=v"(r0), "=v"(r1), "=&s"(sd), "=&s"(cr), "=&s"(e), "=&s"(f)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1)
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "");
    return ((u64)r1 << 32) | r0;
}


// x * 2^(12 e) mod p for the radix-16 butterflies (e = 1..7; canonical in, canonical out, bit-identical to gl_mul_pow2<12 e>).
// y = x << r as three words (r = 12 e mod 32), placed at word offset q = 12 e div 32, folded with 2^64 == EPS, 2^96 == -1,
// 2^128 == -2^32 by the same borrow / carry corrections as gl_mul1 (8 VALU for q = 0 and q = 2, 12 for q = 1; hipcc's
// gl_mul_pow2 is 13).  Scratch: v116..v121.  s_nop pads the carry hazards of the single chain.
#define GL_SHL_Q0(R, RC)                                                                                     \
    asm("v_lshlrev_b32 " GL_V0 ", " #R ", %3\n\t"            /* y0 */                                             \
        "v_alignbit_b32 " GL_V1 ", %4, %3, " #RC "\n\t"      /* y1 */                                             \
        "v_lshrrev_b32 " GL_V2 ", " #RC ", %4\n\t"           /* y2 */                                             \
        "v_mad_u64_u32 " GL_P0 ", %2, " GL_V2 ", -1, " GL_P0 "\n\t" /* T = y2*EPS + (y1:y0)     -> g */         \
        "v_add_co_u32 " GL_V2 ", %5, " GL_V0 ", -1\n\t"           /* u = T + EPS                       -> h */         \
        "s_nop 1\n\t"                                                                                        \
        "v_addc_co_u32 " GL_V3 ", %5, " GL_V1 ", 0, %5\n\t"                                                            \
        "s_nop 1\n\t"                                                                                        \
        "s_or_b64 %2, %2, %5\n\t"                                                                            \
        "v_cndmask_b32 %0, " GL_V0 ", " GL_V2 ", %2\n\t"                                                               \
        "v_cndmask_b32 %1, " GL_V1 ", " GL_V3 ", %2"                                                                   \
        : "=v"(r0), "=v"(r1), "=&s"(g), "+v"(x0), "+v"(x1), "=&s"(h)                                         \
        :                                                                                                    \
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "")
#define GL_SHL_Q1(R, RC)                                                                                     \
    asm("v_lshlrev_b32 " GL_V1 ", " #R ", %3\n\t"            /* L1 = y0 */                                        \
        "v_alignbit_b32 " GL_V4 ", %4, %3, " #RC "\n\t"      /* L2 = y1 */                                        \
        "v_lshrrev_b32 " GL_V5 ", " #RC ", %4\n\t"           /* L3 = y2 */                                        \
        "v_sub_co_u32 " GL_V0 ", %2, 0, " GL_V5 "\n\t"            /* t0 = 0 - L3                      -> borrow */     \
        "s_nop 1\n\t"                                                                                        \
        "v_subbrev_co_u32 " GL_V1 ", %2, 0, " GL_V1 ", %2\n\t"    /* t1 = L1 - borrow                 -> borrow */     \
        "s_nop 1\n\t"                                                                                        \
        "v_addc_co_u32 " GL_V0 ", %5, " GL_V0 ", 0, %2\n\t"       /* -2^64 == -EPS: t0 += 1           -> f */          \
        "s_nop 1\n\t"                                                                                        \
        "s_andn2_b64 %2, %2, %5\n\t"                                                                         \
        "v_subbrev_co_u32 " GL_V1 ", %5, 0, " GL_V1 ", %2\n\t"    /*                 t1 -= borrow & ~f */              \
        "v_mad_u64_u32 " GL_P0 ", %2, " GL_V4 ", -1, " GL_P0 "\n\t" /* T = L2*EPS + t            -> g */        \
        "v_add_co_u32 " GL_V2 ", %5, " GL_V0 ", -1\n\t"                                                                \
        "s_nop 1\n\t"                                                                                        \
        "v_addc_co_u32 " GL_V3 ", %5, " GL_V1 ", 0, %5\n\t"                                                            \
        "s_nop 1\n\t"                                                                                        \
        "s_or_b64 %2, %2, %5\n\t"                                                                            \
        "v_cndmask_b32 %0, " GL_V0 ", " GL_V2 ", %2\n\t"                                                               \
        "v_cndmask_b32 %1, " GL_V1 ", " GL_V3 ", %2"                                                                   \
        : "=v"(r0), "=v"(r1), "=&s"(g), "+v"(x0), "+v"(x1), "=&s"(h)                                         \
        :                                                                                                    \
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V3 "", "" GL_V4 "", "" GL_V5 "")
#define GL_SHL_Q2(R, RC)                                                                                     \
    asm("v_lshlrev_b32 " GL_V2 ", " #R ", %3\n\t"            /* y0 */                                             \
        "v_alignbit_b32 " GL_V4 ", %4, %3, " #RC "\n\t"      /* y1 */                                             \
        "v_lshrrev_b32 " GL_V5 ", " #RC ", %4\n\t"           /* y2 */                                             \
        "v_mad_u64_u32 " GL_P0 ", %2, " GL_V2 ", -1, 0\n\t" /* T = y0*EPS  (< p) */                              \
        "v_sub_co_u32 " GL_V0 ", %2, " GL_V0 ", " GL_V4 "\n\t"         /* T - (y2:y1)                      -> borrow */     \
        "s_nop 1\n\t"                                                                                        \
        "v_subb_co_u32 " GL_V1 ", %2, " GL_V1 ", " GL_V5 ", %2\n\t"                                                         \
        "s_nop 1\n\t"                                                                                        \
        "v_addc_co_u32 %0, %5, " GL_V0 ", 0, %2\n\t"         /* negative: + p == - EPS (mod 2^64) */              \
        "s_nop 1\n\t"                                                                                        \
        "s_andn2_b64 %2, %2, %5\n\t"                                                                         \
        "v_subbrev_co_u32 %1, %5, 0, " GL_V1 ", %2"                                                               \
        : "=v"(r0), "=v"(r1), "=&s"(g), "+v"(x0), "+v"(x1), "=&s"(h)                                         \
        :                                                                                                    \
        : "scc", "" GL_V0 "", "" GL_V1 "", "" GL_V2 "", "" GL_V4 "", "" GL_V5 "")

template <int E>
__device__ __forceinline__ u64 gl_shl12(u64 x) {
    static_assert(E >= 0 && E <= 7, "e in 0..7");
    if constexpr (E == 0) return x;
    u32 x0 = (u32)x, x1 = (u32)(x >> 32), r0, r1;
    u64 g, h;
    if constexpr (E == 1) GL_SHL_Q0(12, 20);
    else if constexpr (E == 2) GL_SHL_Q0(24, 8);
    else if constexpr (E == 3) GL_SHL_Q1(4, 28);
    else if constexpr (E == 4) GL_SHL_Q1(16, 16);
    else if constexpr (E == 5) GL_SHL_Q1(28, 4);
    else if constexpr (E == 6) GL_SHL_Q2(8, 24);
    else GL_SHL_Q2(20, 12);
    return ((u64)r1 << 32) | r0;
}
#endif
