// Poseidon-12 over Goldilocks, linear-hash leaves and binary Merkle trees.
//
// No reference counterpart in /root/reference (hashing happens inside the external prover service
// that src/prover/provider.rs:358-377 calls).  Schedule ARK -> S-box(x^7) -> MDS, 4 full + 22 partial + 4 full
// rounds, tables injected through zp_set_constants.  The field values are the textbook's; on the default matrix the
// throughput kernels walk the partial rounds three at a time (partial3_default: three S-boxes + ONE matrix product,
// round 6) and compute only the digest rows of the last product where only a digest is read.
//
// Mapping: one lane = one permutation, the 12-element state lives in 24 VGPRs; round constants and
// the MDS matrix are wave-uniform and are fetched with scalar loads.  The leaf kernel walks the
// column-major matrix with lane = row, so every column read is a coalesced 512-byte run per wave.
#include <hip/hip_runtime.h>

#include <cstring>

#include "ctx.hpp"
// the Poseidon kernels fit 64 VGPRs (8 waves per SIMD) with the literal-table MDS: low scratch window
#define GL_ASM_SCRATCH_BASE 52
#include "gl_asm.hpp"

namespace {

// x^7 (any u64 in, weak out): the hand-scheduled 15-instruction weak products of gl_asm.hpp -- hipcc's gl_mul_weak is ~28 instructions,
// a third of them moves that build zero-extended 64-bit addends
__device__ __forceinline__ u64 sbox7(u64 x) {
    const u64 x2 = gl_mul1w(x, x);
    u64 x4 = x2, x3 = x2;
    gl_mul2w(x4, x2, x3, x);
    return gl_mul1w(x3, x4);
}
// two S-boxes at once: every product has an independent partner, no padding nops
__device__ __forceinline__ void sbox7x2(u64 &a, u64 &b) {
    u64 a2 = a, b2 = b;
    gl_mul2w(a2, a, b2, b);
    u64 a4 = a2, b4 = b2;
    gl_mul2w(a4, a2, b4, b2);
    gl_mul2w(a, a2, b, b2);        // a^3, b^3
    gl_mul2w(a, a4, b, b4);        // a^7, b^7
}

// default MDS (eigen_zeth_amd/poseidon_constants.py): circulant [17,15,41,16,2,28,13,13,39,18,34,20] + diag [8,0,..]
// coefficient of in[j] in out[i] is circ[(j-i) mod 12] (+8 at i=j=0).  As compile-time literals the
// entries are inline constants of v_mad_u64_u32: no SGPRs, no scalar loads.
__host__ __device__ __forceinline__ constexpr u32 def_mds(int i, int j) {
    constexpr u32 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return circ[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
}

// s <- M*s + c   (c = the NEXT round's constants, or none): out[i] = sum_j m[i][j]*s[j] + c[i].
// m entries < 2^28 and s any u64: the two 64-bit partial sums over 32-bit halves cannot overflow
// (12 * 2^28 * 2^32 + 2^32 < 2^64).  Result weak.
// DEFMDS=false: injected matrix, staged once per workgroup in LDS (keeping 144 entries in SGPRs
// spills them to VGPR lanes; LDS reads of a wave-uniform address are broadcasts).
#include "poseidon_mds_asm.inc"   // mds_ark_default_asm: the default matrix as mad chains, 24 + 4 instructions per row

template <bool ADDC, bool DEFMDS>
__device__ __forceinline__ void mds_ark(u64 *s, const u32 *__restrict__ mds, const u64 *__restrict__ c) {
    if constexpr (DEFMDS) {
        mds_ark_default_asm<ADDC>(s, c);
        return;
    }
    u64 o[12];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        u64 alo = ADDC ? (u64)(u32)c[i] : 0ULL, ahi = ADDC ? (c[i] >> 32) : 0ULL;
        const u32 *mrow = mds + i * 12;  // DEFMDS=false: mds points at the workgroup's LDS copy
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u32 m = DEFMDS ? def_mds(i, j) : mrow[j];
            alo += (u64)m * (u32)s[j];
            ahi += (u64)m * (u32)(s[j] >> 32);
        }
        const u64 mid = (alo >> 32) + ahi;  // value = (u32)alo + mid * 2^32
        o[i] = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = o[i];
}

// rows 0..3 of the default matrix product, no constants (the last round of a permutation whose caller reads a digest only)
__device__ __forceinline__ void mds_last4_default(u64 *s) {
    u32 lo[12], hi[12], o0[4], o1[4];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)s[j]; hi[j] = (u32)(s[j] >> 32); }
    mds_rows_0_1(lo, hi, 0ULL, 0ULL, 0ULL, 0ULL, o0[0], o1[0], o0[1], o1[1]);
    mds_rows_2_3(lo, hi, 0ULL, 0ULL, 0ULL, 0ULL, o0[2], o1[2], o0[3], o1[3]);
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = ((u64)o1[j] << 32) | o0[j];
}

// ---- partial rounds THREE AT A TIME (default matrix; round 6).  A partial round is linear except for ONE S-box, so three of them are
// three S-boxes and ONE matrix product instead of three.  With t = the state after the first S-box (y0 in element 0), Z = "clear
// element 0", c1, c2, c3 the constants added after rounds r, r+1, r+2:
//     x1 = (M t)_0 + c1_0                                        y1 = x1^7
//     x2 = (M Z M t)_0 + M_00 y1 + (M Z c1 + c2)_0               y2 = x2^7
//     state entering round r+3 = (M Z)^2 M t + (M Z M e0) y1 + (M e0) y2 + [(M Z)^2 c1 + M Z c2 + c3]
// All matrices are products of M and M-with-column-0-cleared: non-negative integers below 2^21 (M^3 < 1 525 685), compile-time tables
// that the mads take from SGPRs; the three constant terms (14 words per block) are made on the host whenever a table is installed and sit
// behind the 360 round constants.  12 + 13 + 12 * 14 = 193 row terms per three rounds instead of 3 * 144: the 22 partial rounds cost
// 7 * (386 mads + 14 reductions) + one textbook round instead of 22 * (288 + 12).  The same field values as the textbook schedule
// (the oracle runs the textbook; every digest identical).  ZP_POSEIDON_BLOCK3=0 builds the textbook loop for A/B.
#ifndef ZP_POSEIDON_BLOCK3
#define ZP_POSEIDON_BLOCK3 1
#endif
#define ZP_POSEIDON_PK_BLOCKS 7          // partial rounds 4..24 in blocks of three; round 25 stays textbook
#define ZP_POSEIDON_PK_WORDS 14          // k1, k2, K3[12]
struct P3Tab {
    u32 r1[12];       // row 0 of M
    u32 r2[12];       // row 0 of M Z M
    u32 a3[12][12];   // (M Z)^2 M
    u32 u[12];        // column 0 of M Z M
    u32 v[12];        // column 0 of M
    u32 mz[12][12];   // M Z (host: the constant terms)
};
__host__ __device__ constexpr P3Tab p3_tab() {
    P3Tab t{};
    u32 m[12][12] = {}, mz[12][12] = {}, a2[12][12] = {};
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            m[i][j] = def_mds(i, j);
            mz[i][j] = j == 0 ? 0u : def_mds(i, j);
        }
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            u32 acc = 0;
            for (int k = 0; k < 12; k++) acc += mz[i][k] * m[k][j];
            a2[i][j] = acc;
        }
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++) {
            u32 acc = 0;
            for (int k = 0; k < 12; k++) acc += mz[i][k] * a2[k][j];
            t.a3[i][j] = acc;
            t.mz[i][j] = mz[i][j];
        }
    for (int j = 0; j < 12; j++) {
        t.r1[j] = m[0][j];
        t.r2[j] = a2[0][j];
        t.u[j] = a2[j][0];
        t.v[j] = m[j][0];
    }
    return t;
}
// alo + ahi 2^32, both < 2^58  ->  weak.  The tail of poseidon_mds_asm.inc's rows: T = alo + (ahi >> 32) EPS (no overflow: < 2^59), then
// hi(T) + lo(ahi) with 2^64 == EPS folded back (carry c: lo -= 1, and hi += 1 unless that borrowed) -- four VALU instructions
__device__ __forceinline__ u64 p3_reduce(u64 alo, u64 ahi) {
    const u64 t = (u64)(u32)(ahi >> 32) * 0xFFFFFFFFu + alo;
    const u32 tl = (u32)t, th = (u32)(t >> 32), b0 = (u32)ahi;
    u32 r0, r1;
    u64 c, d;
    asm("v_add_co_u32 %1, %2, %5, %6\n\t"
        "s_nop 0\n\t"
        "v_subbrev_co_u32 %0, %3, 0, %4, %2\n\t"
        "s_nop 0\n\t"
        "s_andn2_b64 %2, %2, %3\n\t"
        "v_addc_co_u32 %1, %3, %1, 0, %2"
        : "=&v"(r0), "=&v"(r1), "=&s"(c), "=&s"(d)
        : "v"(tl), "v"(th), "v"(b0)
        : "scc");
    return ((u64)r1 << 32) | r0;
}
// s: the state entering partial round r (its constants added); pk: the block's 14 constant words.  Leaves the state entering round r + 3.
__device__ __forceinline__ void partial3_default(u64 *s, const u64 *__restrict__ pk) {
    constexpr P3Tab T = p3_tab();
    s[0] = sbox7(s[0]);
    u32 lo[12], hi[12];
#pragma unroll
    for (int j = 0; j < 12; j++) { lo[j] = (u32)s[j]; hi[j] = (u32)(s[j] >> 32); }
    u64 alo = (u64)(u32)pk[0], ahi = pk[0] >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) { alo += (u64)T.r1[j] * lo[j]; ahi += (u64)T.r1[j] * hi[j]; }
    const u64 y1 = sbox7(p3_reduce(alo, ahi));
    const u32 y1l = (u32)y1, y1h = (u32)(y1 >> 32);
    alo = (u64)(u32)pk[1];
    ahi = pk[1] >> 32;
#pragma unroll
    for (int j = 0; j < 12; j++) { alo += (u64)T.r2[j] * lo[j]; ahi += (u64)T.r2[j] * hi[j]; }
    alo += (u64)T.v[0] * y1l;
    ahi += (u64)T.v[0] * y1h;
    const u64 y2 = sbox7(p3_reduce(alo, ahi));
    const u32 y2l = (u32)y2, y2h = (u32)(y2 >> 32);
#pragma unroll
    for (int i = 0; i < 12; i++) {
        alo = (u64)(u32)pk[2 + i];
        ahi = pk[2 + i] >> 32;
#pragma unroll
        for (int j = 0; j < 12; j++) { alo += (u64)T.a3[i][j] * lo[j]; ahi += (u64)T.a3[i][j] * hi[j]; }
        alo += (u64)T.u[i] * y1l;
        ahi += (u64)T.u[i] * y1h;
        alo += (u64)T.v[i] * y2l;
        ahi += (u64)T.v[i] * y2h;
        s[i] = p3_reduce(alo, ahi);
    }
}

// textbook schedule ARK -> S-box -> MDS, with each round's ARK folded into the previous round's
// MDS accumulators; state is weak between rounds and canonicalised once at the end.
// OUT4: the caller reads s[0..4) alone (a digest: Merkle leaves and nodes, the grinding hash) -- four rows of the last matrix product
// instead of twelve; s[4..12) are left stale
template <bool DEFMDS, bool OUT4 = false>
__device__ __forceinline__ void poseidon_perm(u64 *s, const u64 *__restrict__ rc, const u32 *__restrict__ mds) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_add_weak(s[i], rc[i]);
    if constexpr (DEFMDS && ZP_POSEIDON_BLOCK3) {
#pragma unroll 1
        for (int r = 0; r < 4; r++) {
#pragma unroll
            for (int i = 0; i < 12; i += 2) sbox7x2(s[i], s[i + 1]);
            mds_ark<true, true>(s, mds, rc + (r + 1) * 12);
        }
#pragma unroll 1
        for (int b = 0; b < ZP_POSEIDON_PK_BLOCKS; b++) partial3_default(s, rc + 360 + b * ZP_POSEIDON_PK_WORDS);
        s[0] = sbox7(s[0]);                                   // round 25
        mds_ark<true, true>(s, mds, rc + 26 * 12);
#pragma unroll 1
        for (int r = 26; r < 29; r++) {
#pragma unroll
            for (int i = 0; i < 12; i += 2) sbox7x2(s[i], s[i + 1]);
            mds_ark<true, true>(s, mds, rc + (r + 1) * 12);
        }
#pragma unroll
        for (int i = 0; i < 12; i += 2) sbox7x2(s[i], s[i + 1]);
        if constexpr (OUT4) {
            mds_last4_default(s);
#pragma unroll
            for (int i = 0; i < 4; i++) s[i] = gl_canon(s[i]);
        } else {
            mds_ark<false, true>(s, mds, rc);
#pragma unroll
            for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
        }
        return;
    }
#pragma unroll 1
    for (int r = 0; r < 29; r++) {
        if (r < 4 || r >= 26) {
#pragma unroll
            for (int i = 0; i < 12; i += 2) sbox7x2(s[i], s[i + 1]);
        } else {
            s[0] = sbox7(s[0]);
        }
        mds_ark<true, DEFMDS>(s, mds, rc + (r + 1) * 12);
    }
#pragma unroll
    for (int i = 0; i < 12; i += 2) sbox7x2(s[i], s[i + 1]);
    mds_ark<false, DEFMDS>(s, mds, rc);
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = gl_canon(s[i]);
}

template <bool DEFMDS>
__device__ __forceinline__ const u32 *stage_mds(const u32 *mds, u32 *smds) {
    if constexpr (DEFMDS) {
        return mds;
    } else {
        if (threadIdx.x < 144) smds[threadIdx.x] = mds[threadIdx.x];
        __syncthreads();
        return smds;
    }
}

template <bool DEFMDS>
__global__ void __launch_bounds__(256) poseidon_perm_kernel(u64 *states, size_t count, const u64 *rc, const u32 *mds) {
    __shared__ u32 smds[DEFMDS ? 1 : 144];
    mds = stage_mds<DEFMDS>(mds, smds);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = states[i * 12 + j];
    poseidon_perm<DEFMDS>(s, rc, mds);
#pragma unroll
    for (int j = 0; j < 12; j++) states[i * 12 + j] = s[j];
}

// Latency form for a handful of permutations (Fiat-Shamir transcript: one state at a time).  One lane per
// state WORD instead of one lane per state: 12 lanes share a permutation through LDS, so a round is one
// S-box + one 12-term row sum per lane (~150 dependent instructions) instead of ~1600 on a single lane.
__global__ void __launch_bounds__(64) poseidon_perm_small_kernel(u64 *states, int count, const u64 *rc, const u32 *mds) {
    __shared__ u64 sh[5][12];
    const int lane = threadIdx.x, q = lane / 12, e = lane % 12;
    const int perm = blockIdx.x * 5 + q;
    const bool on = q < 5 && perm < count;
    u64 s = on ? states[(size_t)perm * 12 + e] : 0ULL;
    for (int r = 0; r < 30; r++) {
        s = gl_add_weak(s, rc[r * 12 + e]);
        if (r < 4 || r >= 26 || e == 0) s = sbox7(s);
        if (on) sh[q][e] = s;
        __syncthreads();
        u64 alo = 0, ahi = 0;
        if (on) {
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const u64 v = sh[q][j];
                const u32 m = mds[e * 12 + j];
                alo += (u64)m * (u32)v;
                ahi += (u64)m * (u32)(v >> 32);
            }
        }
        __syncthreads();
        const u64 mid = (alo >> 32) + ahi;
        s = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
    }
    if (on) states[(size_t)perm * 12 + e] = gl_canon(s);
}

// one permutation of the state a wave holds in its lanes 0..11 (the single-wave kernels: transcript steps, recursion-witness walks)
// (the round constants come from LDS -- rcs: all 360, loaded once per kernel by wave12_tables -- and the lane's matrix row from registers: a
// permutation of these single-wave kernels is a chain of 30 dependent rounds, and a global load of a constant sat in every link)
__device__ __forceinline__ void wave12_tables(int e, bool on, const u64 *rc, const u32 *mds, u64 *rcs, u32 *row) {
    for (int i = threadIdx.x; i < 360; i += 64) rcs[i] = rc[i];
#pragma unroll
    for (int j = 0; j < 12; j++) row[j] = mds[(on ? e : 0) * 12 + j];
    __syncthreads();
}
__device__ __forceinline__ u64 wave12_perm(u64 s, int e, bool on, u64 *sh, const u64 *rcs, const u32 *row) {
    for (int r = 0; r < 30; r++) {
        s = gl_add_weak(s, rcs[r * 12 + (on ? e : 0)]);
        if (r < 4 || r >= 26 || e == 0) s = sbox7(s);
        if (on) sh[e] = s;
        __syncthreads();
        u64 alo = 0, ahi = 0;
        if (on) {
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const u64 v = sh[j];
                const u32 m = row[j];
                alo += (u64)m * (u32)v;
                ahi += (u64)m * (u32)(v >> 32);
            }
        }
        __syncthreads();
        const u64 mid = (alo >> 32) + ahi;
        s = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
    }
    return gl_canon(s);
}

// The Fiat-Shamir sponge as ONE launch: buf = [12 state words][nblocks x 8 block words][(1 + extra) x 8 rate words out].
// For every block: the rate (state[0..8)) is overwritten with the block, then one permutation (no block: one permutation);
// then `extra` further permutations, the rate after each of the 1 + extra steps is written out.  Same 12-lanes-per-state form
// as above: a transcript of k permutations costs one host round trip instead of k.
// caps (optional): the capacity (state[8..12)) after EVERY permutation, 4 words each -- with the blocks and the rates that is the
// input state of every permutation of the step, what the verifier AIR's witness needs (stark/verifier_air.py: transcript blocks).
__global__ void __launch_bounds__(64) poseidon_sponge_kernel(u64 *buf, int nblocks, int extra, const u64 *rc, const u32 *mds, u64 *caps) {
    __shared__ u64 sh[12];
    __shared__ u64 rcs[360];
    const int e = threadIdx.x;
    const bool on = e < 12;
    u32 row[12];
    wave12_tables(e, on, rc, mds, rcs, row);
    u64 s = on ? buf[e] : 0ULL;
    u64 *rates = buf + 12 + (size_t)nblocks * 8;
    const int absorb = nblocks > 0 ? nblocks : 1;
    for (int b = 0; b < absorb + extra; b++) {
        if (b < nblocks && e < 8) s = buf[12 + (size_t)b * 8 + e];
        s = wave12_perm(s, e, on, sh, rcs, row);
        if (b >= absorb - 1 && e < 8) rates[(size_t)(b - (absorb - 1)) * 8 + e] = s;
        if (caps && on && e >= 8) caps[(size_t)b * 4 + (e - 8)] = s;
    }
    if (on) buf[e] = s;
}

// The same step with its buffer in PAGE-LOCKED, device-visible host memory (round 5): the host writes state and blocks into the ctx's staging
// buffer, this kernel pulls them into LDS in one coalesced sweep, walks them, and posts state, rates and capacities straight back -- one launch
// and two stream synchronisations per transcript step instead of four launches (copy kernels in and out) and four synchronisations: a step
// went from ~180 us to ~45 us, and a proof has 9-12 of them, a recursion witness ~24.
__global__ void __launch_bounds__(64) poseidon_sponge_pinned_kernel(u64 *buf, int nblocks, int extra, const u64 *rc, const u32 *mds, int want_caps) {
    extern __shared__ u64 stage[];              // [12 + nblocks * 8]
    __shared__ u64 sh[12];
    __shared__ u64 rcs[360];
    const int e = threadIdx.x;
    const bool on = e < 12;
    u32 row[12];
    wave12_tables(e, on, rc, mds, rcs, row);
    const int nin = 12 + nblocks * 8;
    for (int i = e; i < nin; i += 64) stage[i] = buf[i];
    __syncthreads();
    u64 s = on ? stage[e] : 0ULL;
    u64 *rates = buf + nin, *caps = rates + (size_t)(1 + extra) * 8;
    const int absorb = nblocks > 0 ? nblocks : 1;
    for (int b = 0; b < absorb + extra; b++) {
        if (b < nblocks && e < 8) s = stage[12 + b * 8 + e];
        s = wave12_perm(s, e, on, sh, rcs, row);
        if (b >= absorb - 1 && e < 8) rates[(size_t)(b - (absorb - 1)) * 8 + e] = s;
        if (want_caps && on && e >= 8) caps[(size_t)b * 4 + (e - 8)] = s;
    }
    if (on) buf[e] = s;
}

// ---- the hashing walk of a recursion witness on the device (csrc/recursion.hip, round 5): no host round trip per permutation
// Whole Fiat-Shamir transcripts in ONE launch: workgroup c walks the steps [first[c], first[c + 1]) of chain c from the zero state; a step
// overwrites the rate with its block (absorb[s] != 0) or just permutes; out[s] = the 12 words entering the permutation, then the rate after it
__global__ void __launch_bounds__(64) sponge_chains_kernel(const u64 *__restrict__ blocks, const unsigned char *__restrict__ absorb,
                                                           const unsigned int *__restrict__ first, u64 *__restrict__ out, const u64 *rc, const u32 *mds) {
    __shared__ u64 sh[12];
    __shared__ u64 rcs[360];
    const int e = threadIdx.x;
    const bool on = e < 12;
    u32 row[12];
    wave12_tables(e, on, rc, mds, rcs, row);
    u64 s = 0;
    for (unsigned int st = first[blockIdx.x]; st < first[blockIdx.x + 1]; st++) {
        if (absorb[st] && e < 8) s = blocks[(size_t)st * 8 + e];
        if (on) out[(size_t)st * 20 + e] = s;
        s = wave12_perm(s, e, on, sh, rcs, row);
        if (e < 8) out[(size_t)st * 20 + 12 + e] = s;
    }
}

// Every opening of a recursion witness in ONE launch: workgroup o hashes the leaf of opening o (na blocks of 8 values: the linear hash of
// merkle_leaves_kernel -- words 0..3 of a permutation's output are the capacity of the next block) and walks its nd path levels along the bits
// of its index; the 12 words entering every permutation go to inputs[b0 + ...] (the verifier AIR's permutation blocks, in HBM where
// zp_poseidon_trace reads them), the last digest to digests[o].   op: b0, na, nd per opening; vals: mw words per opening, zero padded;
// sib: the siblings of opening o at sib_off[o], 4 words per level
__global__ void __launch_bounds__(64) openings_walk_kernel(const u64 *__restrict__ op, const u64 *__restrict__ vals, u64 mw, const u64 *__restrict__ index,
                                                           const u64 *__restrict__ sib, const u64 *__restrict__ sib_off, u64 *__restrict__ inputs,
                                                           u64 *__restrict__ digests, const u64 *rc, const u32 *mds) {
    __shared__ u64 sh[12];
    __shared__ u64 cur[4];
    __shared__ u64 rcs[360];
    const int e = threadIdx.x;
    const bool on = e < 12;
    u32 row[12];
    wave12_tables(e, on, rc, mds, rcs, row);
    const u64 o = blockIdx.x, b0 = op[3 * o], na = op[3 * o + 1], nd = op[3 * o + 2];
    const u64 *v = vals + o * mw;
    u64 s = 0;
    if (e < 4) cur[e] = v[e];                               // an unhashed leaf (<= 4 values) is its own digest, zero padded
    for (u64 j = 0; j < na; j++) {
        if (e < 8) s = 8 * j + e < mw ? v[8 * j + e] : 0ULL;
        if (on) inputs[(b0 + j) * 12 + e] = s;
        s = wave12_perm(s, e, on, sh, rcs, row);
        if (on) sh[e] = s;
        __syncthreads();
        if (e >= 8 && on) s = sh[e - 8];                    // the digest so far is the next block's capacity
        if (e < 4) cur[e] = s;
        __syncthreads();
    }
    const u64 idx = index[o];
    const u64 *sb = sib + sib_off[o];
    __syncthreads();
    for (u64 lv = 0; lv < nd; lv++) {
        const bool bit = (idx >> lv) & 1;
        if (on) {
            if (e < 4) s = bit ? sb[lv * 4 + e] : cur[e];
            else if (e < 8) s = bit ? cur[e - 4] : sb[lv * 4 + (e - 4)];
            else s = 0;
            inputs[(b0 + na + lv) * 12 + e] = s;
        }
        __syncthreads();
        s = wave12_perm(s, e, on, sh, rcs, row);
        if (e < 4) cur[e] = s;
        __syncthreads();
    }
    if (e < 4) digests[o * 4 + e] = cur[e];
}

// proof-of-work grinding (before the query phase of a STARK): lane = candidate nonce base + gid; a hit is a nonce with
// Poseidon(seed[0..3] || nonce || 0^7)[0] >> (64 - bits) == 0; the smallest hit of the batch wins (atomicMin).
template <bool DEFMDS>
__global__ void __launch_bounds__(256) pow_grind_kernel(const u64 *__restrict__ seed4, int bits, u64 base, u64 *best,
                                                       const u64 *rc, const u32 *mds) {
    __shared__ u32 smds[DEFMDS ? 1 : 144];
    mds = stage_mds<DEFMDS>(mds, smds);
    const u64 nonce = base + (u64)blockIdx.x * 256 + threadIdx.x;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 4; j++) s[j] = seed4[j];
    s[4] = nonce;
#pragma unroll
    for (int j = 5; j < 12; j++) s[j] = 0;
    poseidon_perm<DEFMDS, true>(s, rc, mds);
    if ((s[0] >> (64 - bits)) == 0) atomicMin((unsigned long long *)best, (unsigned long long)nonce);
}

// Round-by-round trace of permutations: the witness of a Poseidon AIR (stark/verifier_air.py).  lane = permutation k, block of 32
// rows: states[e][32 k + r] = element e of the state BEFORE round r (r < 30), the output on rows 30 and 31;
// cubes[e][32 k + r] = (state + round constant)^3 (rows 30, 31: state^3) -- the helper column that keeps x^7 at degree 3.
// Canonical values throughout (they are committed).  Column stride in elements.
__global__ void __launch_bounds__(128) poseidon_trace_kernel(const u64 *__restrict__ inputs, size_t count, u64 *__restrict__ states,
                                                            u64 *__restrict__ cubes, size_t stride, const u64 *__restrict__ rc,
                                                            const u32 *__restrict__ mds) {
    __shared__ u32 smds[144];
    if (threadIdx.x < 144 - 128) smds[128 + threadIdx.x] = mds[128 + threadIdx.x];
    smds[threadIdx.x] = mds[threadIdx.x];
    __syncthreads();
    const size_t k = (size_t)blockIdx.x * 128 + threadIdx.x;
    if (k >= count) return;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = inputs[k * 12 + j];
    const size_t row0 = 32 * k;
#pragma unroll 1
    for (int r = 0; r < 32; r++) {
        const bool round = r < 30;
        const bool full = r < 4 || r >= 26;
        u64 y[12];
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u64 x = round ? gl_add(s[j], rc[r * 12 + j]) : s[j];
            const u64 c = gl_mul(gl_mul(x, x), x);
            states[(size_t)j * stride + row0 + r] = s[j];
            cubes[(size_t)j * stride + row0 + r] = c;
            y[j] = (full || j == 0) ? gl_mul(gl_mul(c, c), x) : x;
        }
        if (!round) continue;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            u64 alo = 0, ahi = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const u32 m = smds[i * 12 + j];
                alo += (u64)m * (u32)y[j];
                ahi += (u64)m * (u32)(y[j] >> 32);
            }
            const u64 mid = (alo >> 32) + ahi;
            s[i] = gl_canon(gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u));
        }
    }
}

// leaf i = linear hash of (cols[0][i], cols[1][i], ... cols[W-1][i]);  lane = row
template <bool DEFMDS>
__global__ void __launch_bounds__(256) merkle_leaves_kernel(const u64 *__restrict__ cols, size_t M, int W,
                                                           u64 *__restrict__ tree, const u64 *rc, const u32 *mds) {
    __shared__ u32 smds[DEFMDS ? 1 : 144];
    mds = stage_mds<DEFMDS>(mds, smds);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    u64 s[12];
    if (W <= 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) tree[i * 4 + j] = j < W ? cols[(size_t)j * M + i] : 0ULL;
        return;
    }
#pragma unroll
    for (int j = 8; j < 12; j++) s[j] = 0;
    for (int off = 0; off < W; off += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = (off + j < W) ? cols[(size_t)(off + j) * M + i] : 0ULL;
        poseidon_perm<DEFMDS, true>(s, rc, mds);
#pragma unroll
        for (int j = 0; j < 4; j++) s[8 + j] = s[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) tree[i * 4 + j] = s[8 + j];
}

// Small trees (FRI layers below 2^14 leaves): 12 lanes per leaf, like merkle_subtree_kernel -- a leaf of 24 values is
// three dependent permutations, 0.33 ms on a single lane while the chip idles, ~50 us with the state spread over lanes.
// One permutation of the 64 states a workgroup holds, lane = (state, element); sh is the exchange buffer.
__device__ __forceinline__ u64 coop_perm(u64 s, int node, int e, u64 (*sh)[12], const u64 *rc, const u32 *mds) {
    for (int r = 0; r < 30; r++) {
        s = gl_add_weak(s, rc[r * 12 + e]);
        if (r < 4 || r >= 26 || e == 0) s = sbox7(s);
        sh[node][e] = s;
        __syncthreads();
        u64 alo = 0, ahi = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            const u64 v = sh[node][j];
            const u32 m = mds[e * 12 + j];
            alo += (u64)m * (u32)v;
            ahi += (u64)m * (u32)(v >> 32);
        }
        __syncthreads();
        const u64 mid = (alo >> 32) + ahi;
        s = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
    }
    return gl_canon(s);
}
// stride_e / stride_i: element (row i, position k) sits at src[k * stride_e + i * stride_i]  (columns: M, 1; rows: 1, len)
__global__ void __launch_bounds__(768) merkle_leaves_coop_kernel(const u64 *__restrict__ src, size_t M, size_t len, size_t stride_e,
                                                                size_t stride_i, u64 *__restrict__ tree, const u64 *rc,
                                                                const u32 *mds) {
    __shared__ u64 sh[64][12];
    const int node = threadIdx.x / 12, e = threadIdx.x % 12;
    const size_t i = (size_t)blockIdx.x * 64 + node;
    const bool on = i < M;                       // every lane runs the barriers; idle leaves compute on zeros
    if (len <= 4) {
        if (on && e < 4) tree[i * 4 + e] = (size_t)e < len ? src[(size_t)e * stride_e + i * stride_i] : 0ULL;
        return;
    }
    u64 s = 0;
    for (size_t off = 0; off < len; off += 8) {
        if (e < 8) s = (on && off + e < len) ? src[(off + e) * stride_e + i * stride_i] : 0ULL;
        s = coop_perm(s, node, e, sh, rc, mds);
        // capacity of the next block = the first four outputs
        sh[node][e] = s;
        __syncthreads();
        const u64 cap = sh[node][e & 3];
        __syncthreads();
        if (e >= 8) s = cap;
        else if (off + 8 >= len && e < 4 && on) tree[i * 4 + e] = s;
    }
}

// leaves given as M contiguous rows of `len` elements
template <bool DEFMDS>
__global__ void __launch_bounds__(256) merkle_leaves_rows_kernel(const u64 *__restrict__ rows, size_t M, size_t len,
                                                                u64 *__restrict__ tree, const u64 *rc, const u32 *mds) {
    __shared__ u32 smds[DEFMDS ? 1 : 144];
    mds = stage_mds<DEFMDS>(mds, smds);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const u64 *row = rows + i * len;
    u64 s[12];
    if (len <= 4) {
#pragma unroll
        for (int j = 0; j < 4; j++) tree[i * 4 + j] = (size_t)j < len ? row[j] : 0ULL;
        return;
    }
#pragma unroll
    for (int j = 8; j < 12; j++) s[j] = 0;
    for (size_t off = 0; off < len; off += 8) {
#pragma unroll
        for (int j = 0; j < 8; j++) s[j] = (off + j < len) ? row[off + j] : 0ULL;
        poseidon_perm<DEFMDS, true>(s, rc, mds);
#pragma unroll
        for (int j = 0; j < 4; j++) s[8 + j] = s[j];
    }
#pragma unroll
    for (int j = 0; j < 4; j++) tree[i * 4 + j] = s[8 + j];
}

// one tree level: node i = P(child[2i] || child[2i+1] || 0^4)[0..4]
template <bool DEFMDS>
__global__ void __launch_bounds__(256) merkle_level_kernel(const u64 *__restrict__ prev, u64 *__restrict__ next,
                                                          size_t half, const u64 *rc, const u32 *mds) {
    __shared__ u32 smds[DEFMDS ? 1 : 144];
    mds = stage_mds<DEFMDS>(mds, smds);
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= half) return;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 8; j++) s[j] = prev[i * 8 + j];
#pragma unroll
    for (int j = 8; j < 12; j++) s[j] = 0;
    poseidon_perm<DEFMDS, true>(s, rc, mds);
#pragma unroll
    for (int j = 0; j < 4; j++) next[i * 4 + j] = s[j];
}

// upper part of the tree, up to 7 levels per launch, 12 lanes per node: workgroup b reduces nodes
// [128b, 128b+128) of the level at `prev` (cnt nodes, a power of two) through nlev levels.  The lane-per-node kernel
// below ~2^15 nodes is latency-bound (one permutation takes a single lane ~0.1 ms); spreading a state over 12 lanes
// brings a level to ~30 us, and fusing the levels of a 128-node subtree removes the launches in between.
__global__ void __launch_bounds__(768) merkle_subtree_kernel(u64 *prev, size_t cnt, int nlev, const u64 *rc, const u32 *mds) {
    __shared__ u64 sh[64][12];
    const int node = threadIdx.x / 12, e = threadIdx.x % 12;
    size_t first = (size_t)blockIdx.x * 128;          // first node of this workgroup in the current level
    size_t width = cnt < 128 ? cnt : 128;             // nodes of the current level this workgroup owns
    for (int l = 0; l < nlev; l++) {
        const size_t half = width >> 1;
        u64 *next = prev + cnt * 4;
        const bool on = (size_t)node < half;
        u64 s = (on && e < 8) ? prev[(first + 2 * (size_t)node) * 4 + e] : 0ULL;
        for (int r = 0; r < 30; r++) {
            s = gl_add_weak(s, rc[r * 12 + e]);
            if (r < 4 || r >= 26 || e == 0) s = sbox7(s);
            sh[node][e] = s;
            __syncthreads();
            u64 alo = 0, ahi = 0;
#pragma unroll
            for (int j = 0; j < 12; j++) {
                const u64 v = sh[node][j];
                const u32 m = mds[e * 12 + j];
                alo += (u64)m * (u32)v;
                ahi += (u64)m * (u32)(v >> 32);
            }
            __syncthreads();
            const u64 mid = (alo >> 32) + ahi;
            s = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
        }
        if (on && e < 4) next[((first >> 1) + (size_t)node) * 4 + e] = gl_canon(s);
        __threadfence_block();
        __syncthreads();
        prev = next;
        cnt >>= 1;
        first >>= 1;
        width = half;
    }
}

// The same subtree walk with the state exchanged by WAVE SHUFFLES instead of LDS + two workgroup barriers per round (round 5; knob
// merkle_top_wave): a node's 12 state words sit in 12 lanes of one 16-lane row (four nodes per wave, lanes 12..15 of a row idle), a round's row
// sum gathers them with twelve 64-bit shuffles whose source lane (row base + (e + k) mod 12) and coefficient m[e][(e + k) mod 12] are fixed per
// lane for the whole kernel, and nothing synchronises inside a permutation -- one workgroup barrier per tree LEVEL (the next level reads what
// other waves wrote) instead of sixty.  64 parents per 1024-thread workgroup and level, as the LDS form.  Same values (tests/test_gpu_parity.py
// compares every node of small trees with the oracle under both knob settings).
__global__ void __launch_bounds__(1024) merkle_subtree_wave_kernel(u64 *prev, size_t cnt, int nlev, const u64 *rc, const u32 *mds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int node = wv * 4 + (lane >> 4), e = lane & 15, rowbase = lane & ~15;
    const bool word = e < 12;
    const int ee = word ? e : 11;
    int src[12];
    u32 m[12];
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int j = (ee + k) % 12;
        src[k] = rowbase + j;
        m[k] = word ? mds[ee * 12 + j] : 0u;
    }
    size_t first = (size_t)blockIdx.x * 128;
    size_t width = cnt < 128 ? cnt : 128;
    for (int l = 0; l < nlev; l++) {
        const size_t half = width >> 1;
        u64 *next = prev + cnt * 4;
        const bool on = (size_t)node < half && word;
        u64 s = (on && e < 8) ? prev[(first + 2 * (size_t)node) * 4 + e] : 0ULL;
        for (int r = 0; r < 30; r++) {
            s = gl_add_weak(s, rc[r * 12 + ee]);
            if (r < 4 || r >= 26 || e == 0) s = sbox7(s);
            u64 alo = 0, ahi = 0;
#pragma unroll
            for (int k = 0; k < 12; k++) {
                const u64 v = (u64)__shfl((unsigned long long)s, src[k]);
                alo += (u64)m[k] * (u32)v;
                ahi += (u64)m[k] * (u32)(v >> 32);
            }
            const u64 mid = (alo >> 32) + ahi;
            s = gl_reduce96_weak(((u64)(u32)mid << 32) | (u32)alo, (u32)(mid >> 32), 0u);
        }
        if (on && e < 4) next[((first >> 1) + (size_t)node) * 4 + e] = gl_canon(s);
        __threadfence_block();
        __syncthreads();
        prev = next;
        cnt >>= 1;
        first >>= 1;
        width = half;
    }
}

int32_t tree_levels(zp_ctx *ctx, u64 *tree, size_t M) {
    u64 *prev = tree;
    size_t cnt = M;
    while (cnt > 1) {
        u64 *next = prev + cnt * 4;
        const size_t half = cnt >> 1;
        if (cnt <= ((size_t)1 << (ctx->tune_merkle_coop_log > 0 ? ctx->tune_merkle_coop_log : 15))) {
            int lg = 0;
            while (((size_t)1 << lg) < cnt) lg++;
            const int nlev = lg < 7 ? lg : 7;
            if (ctx->tune_merkle_top_wave)
                hipLaunchKernelGGL(merkle_subtree_wave_kernel, dim3((unsigned)(cnt < 128 ? 1 : cnt / 128)), dim3(1024), 0, ctx->stream, prev, cnt, nlev,
                                   ctx->d_rc, ctx->d_mds);
            else
                hipLaunchKernelGGL(merkle_subtree_kernel, dim3((unsigned)(cnt < 128 ? 1 : cnt / 128)), dim3(768), 0, ctx->stream,
                                   prev, cnt, nlev, ctx->d_rc, ctx->d_mds);
            ZP_HIP(ctx, hipGetLastError());
            for (int l = 0; l < nlev; l++) {
                prev += cnt * 4;
                cnt >>= 1;
            }
            continue;
        }
        if (ctx->mds_is_default)
            hipLaunchKernelGGL(merkle_level_kernel<true>, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, ctx->stream,
                               prev, next, half, ctx->d_rc, ctx->d_mds);
        else
            hipLaunchKernelGGL(merkle_level_kernel<false>, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, ctx->stream,
                               prev, next, half, ctx->d_rc, ctx->d_mds);
        ZP_HIP(ctx, hipGetLastError());
        prev = next;
        cnt = half;
    }
    return ZP_OK;
}

}  // namespace

int32_t zpi_poseidon_sync_tables(zp_ctx *ctx) {
    if (!ctx->poseidon_dirty) return ZP_OK;
    for (int i = 0; i < 144; i++) ZP_ARG(ctx, ctx->h_mds[i] < (1ULL << 28), "MDS entries must be < 2^28");
    if (!ctx->d_rc) ZP_HIP(ctx, hipMalloc((void **)&ctx->d_rc, (360 + ZP_POSEIDON_PK_BLOCKS * ZP_POSEIDON_PK_WORDS) * sizeof(u64)));
    if (!ctx->d_mds) ZP_HIP(ctx, hipMalloc((void **)&ctx->d_mds, 144 * sizeof(u32)));
    u32 m32[144];
    for (int i = 0; i < 144; i++) m32[i] = (u32)ctx->h_mds[i];
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZP_HIP(ctx, hipMemcpy(ctx->d_rc, ctx->h_rc, 360 * sizeof(u64), hipMemcpyHostToDevice));
    ZP_HIP(ctx, hipMemcpy(ctx->d_mds, m32, sizeof(m32), hipMemcpyHostToDevice));
    ctx->mds_is_default = true;
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 12; j++)
            if (ctx->h_mds[i * 12 + j] != def_mds(i, j)) ctx->mds_is_default = false;
    if (ctx->mds_is_default) {
        // constant terms of the three-round blocks (partial3_default): k1 = c1_0, k2 = (M Z c1 + c2)_0, K3 = (M Z)^2 c1 + M Z c2 + c3
        constexpr P3Tab T = p3_tab();
        u64 pk[ZP_POSEIDON_PK_BLOCKS * ZP_POSEIDON_PK_WORDS];
        auto mzv = [&](const u64 *x, u64 *y) {
            for (int i = 0; i < 12; i++) {
                u64 acc = 0;
                for (int j = 1; j < 12; j++) acc = gl_add(acc, gl_mul((u64)T.mz[i][j], x[j]));
                y[i] = acc;
            }
        };
        for (int b = 0; b < ZP_POSEIDON_PK_BLOCKS; b++) {
            const int r = 4 + 3 * b;
            const u64 *c1 = ctx->h_rc + (r + 1) * 12, *c2 = ctx->h_rc + (r + 2) * 12, *c3 = ctx->h_rc + (r + 3) * 12;
            u64 t1[12], t2[12], t3[12];
            mzv(c1, t1);                 // M Z c1
            mzv(t1, t2);                 // (M Z)^2 c1
            mzv(c2, t3);                 // M Z c2
            u64 *o = pk + b * ZP_POSEIDON_PK_WORDS;
            o[0] = gl_canon(c1[0]);
            o[1] = gl_add(t1[0], gl_canon(c2[0]));
            for (int i = 0; i < 12; i++) o[2 + i] = gl_add(gl_add(t2[i], t3[i]), gl_canon(c3[i]));
        }
        ZP_HIP(ctx, hipMemcpy(ctx->d_rc + 360, pk, sizeof(pk), hipMemcpyHostToDevice));
    }
    ctx->poseidon_dirty = false;
    return ZP_OK;
}

// nsteps steps of nchains sponge chains in one launch (sponge_chains_kernel); d_*: device buffers the caller filled, d_out u64[nsteps][20]
int32_t zpi_poseidon_chains(zp_ctx *ctx, const u64 *d_blocks, const unsigned char *d_absorb, const unsigned int *d_first, int nchains, u64 *d_out) {
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    if (nchains <= 0) return ZP_OK;
    hipLaunchKernelGGL(sponge_chains_kernel, dim3((unsigned)nchains), dim3(64), 0, ctx->stream, d_blocks, d_absorb, d_first, d_out, ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

// the leaf hash and the path of `count` openings in one launch (openings_walk_kernel)
int32_t zpi_poseidon_openings_walk(zp_ctx *ctx, const u64 *d_op, const u64 *d_vals, u64 mw, const u64 *d_index, const u64 *d_sib, const u64 *d_sib_off,
                                   size_t count, u64 *d_inputs, u64 *d_digests) {
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    if (!count) return ZP_OK;
    hipLaunchKernelGGL(openings_walk_kernel, dim3((unsigned)count), dim3(64), 0, ctx->stream, d_op, d_vals, mw, d_index, d_sib, d_sib_off, d_inputs, d_digests,
                       ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

extern "C" {

int32_t zp_poseidon_perm(zp_ctx *ctx, uint64_t *d_states, size_t count) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_perm");
    if (count == 0) return ZP_OK;
    ZP_ARG(ctx, d_states != nullptr, "null device pointer");
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    if (count <= 64) {   // latency-bound regime: spread each state over 12 lanes
        hipLaunchKernelGGL(poseidon_perm_small_kernel, dim3((unsigned)((count + 4) / 5)), dim3(64), 0, ctx->stream,
                           (u64 *)d_states, (int)count, ctx->d_rc, ctx->d_mds);
        ZP_HIP(ctx, hipGetLastError());
        return ZP_OK;
    }
    if (ctx->mds_is_default)
        hipLaunchKernelGGL(poseidon_perm_kernel<true>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream,
                           (u64 *)d_states, count, ctx->d_rc, ctx->d_mds);
    else
        hipLaunchKernelGGL(poseidon_perm_kernel<false>, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream,
                           (u64 *)d_states, count, ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

int32_t zp_poseidon_trace(zp_ctx *ctx, const uint64_t *d_inputs, size_t count, uint64_t *d_states, uint64_t *d_cubes, size_t stride) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_trace");
    if (count == 0) return ZP_OK;
    ZP_ARG(ctx, d_inputs && d_states && d_cubes, "null device pointer");
    ZP_ARG(ctx, stride >= 32 * count, "column stride smaller than 32 rows per permutation");
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    hipLaunchKernelGGL(poseidon_trace_kernel, dim3((unsigned)((count + 127) / 128)), dim3(128), 0, ctx->stream, (const u64 *)d_inputs, count,
                       (u64 *)d_states, (u64 *)d_cubes, stride, ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

int32_t zp_poseidon_sponge_caps(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates,
                                uint64_t *h_caps) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_sponge");
    ZP_ARG(ctx, h_state && h_rates && (h_blocks || nblocks == 0), "null pointer");
    ZP_ARG(ctx, nblocks <= 65536 && extra <= 65536, "too many blocks");
    for (int i = 0; i < 12; i++) ZP_ARG(ctx, h_state[i] < GL_P, "state not canonical");
    for (size_t i = 0; i < nblocks * 8; i++) ZP_ARG(ctx, h_blocks[i] < GL_P, "block not canonical");
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    const size_t nin = 12 + nblocks * 8, nout = (1 + extra) * 8, ncap = h_caps ? ((nblocks ? nblocks : 1) + extra) * 4 : 0;
    if (nin * 8 <= 48 * 1024) {                 // the usual case (a transcript step is a few dozen blocks): through the pinned staging buffer
        void *stv = nullptr;
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));       // the staging buffer may still feed an earlier small copy
        ZP_TRY(zpi_pinned(ctx, (nin + nout + ncap) * 8, &stv));
        u64 *st = (u64 *)stv;
        memcpy(st, h_state, 96);
        if (nblocks) memcpy(st + 12, h_blocks, nblocks * 64);
        hipLaunchKernelGGL(poseidon_sponge_pinned_kernel, dim3(1), dim3(64), nin * 8, ctx->stream, st, (int)nblocks, (int)extra, ctx->d_rc, ctx->d_mds,
                           h_caps ? 1 : 0);
        ZP_HIP(ctx, hipGetLastError());
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(h_state, st, 96);
        memcpy(h_rates, st + nin, nout * 8);
        if (h_caps) memcpy(h_caps, st + nin + nout, ncap * 8);
        return ZP_OK;
    }
    u64 *d = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, nin + nout + ncap, &d));
    std::vector<u64> in(nin);
    memcpy(in.data(), h_state, 96);
    if (nblocks) memcpy(in.data() + 12, h_blocks, nblocks * 64);
    ZP_TRY(zpi_h2d_small(ctx, d, in.data(), nin * 8));
    hipLaunchKernelGGL(poseidon_sponge_kernel, dim3(1), dim3(64), 0, ctx->stream, d, (int)nblocks, (int)extra, ctx->d_rc, ctx->d_mds,
                       h_caps ? d + nin + nout : (u64 *)nullptr);
    ZP_HIP(ctx, hipGetLastError());
    std::vector<u64> out(12 + nout + ncap);
    // state and rates are not adjacent (the blocks sit between them): two small copies
    ZP_TRY(zpi_d2h_small(ctx, out.data(), d, 96));
    ZP_TRY(zpi_d2h_small(ctx, out.data() + 12, d + nin, (nout + ncap) * 8));
    memcpy(h_state, out.data(), 96);
    memcpy(h_rates, out.data() + 12, nout * 8);
    if (h_caps) memcpy(h_caps, out.data() + 12 + nout, ncap * 8);
    return ZP_OK;
}

int32_t zp_poseidon_sponge(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates) {
    return zp_poseidon_sponge_caps(ctx, h_state, h_blocks, nblocks, extra, h_rates, nullptr);
}

int32_t zp_pow_grind(zp_ctx *ctx, const uint64_t *h_seed4, int32_t bits, uint64_t *h_nonce) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "pow_grind");
    ZP_ARG(ctx, h_seed4 && h_nonce, "null pointer");
    ZP_ARG(ctx, bits >= 0 && bits <= 40, "bits must be in [0,40]");
    for (int i = 0; i < 4; i++) ZP_ARG(ctx, h_seed4[i] < GL_P, "seed not canonical");
    if (bits == 0) { *h_nonce = 0; return ZP_OK; }
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    u64 *d;   // [0..3] seed, [4] best nonce of the batch
    ZP_TRY(zpi_scratch(ctx, 3, 8, &d));
    u64 h[5] = {h_seed4[0], h_seed4[1], h_seed4[2], h_seed4[3], ~0ULL};
    ZP_TRY(zpi_h2d_small(ctx, d, h, sizeof(h)));
    const u64 batch = 1ULL << (bits >= 16 ? 20 : bits + 4);   // expected hits per batch: 16 (fewer for bits >= 16)
    for (u64 base = 0;; base += batch) {
        ZP_ARG(ctx, base < (1ULL << 50), "no proof-of-work nonce below 2^50");
        if (ctx->mds_is_default)
            hipLaunchKernelGGL(pow_grind_kernel<true>, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, d, (int)bits, base, d + 4,
                               ctx->d_rc, ctx->d_mds);
        else
            hipLaunchKernelGGL(pow_grind_kernel<false>, dim3((unsigned)(batch / 256)), dim3(256), 0, ctx->stream, d, (int)bits, base, d + 4,
                               ctx->d_rc, ctx->d_mds);
        ZP_HIP(ctx, hipGetLastError());
        u64 best;
        ZP_TRY(zpi_d2h_small(ctx, &best, d + 4, sizeof(best)));
        if (best != ~0ULL) { *h_nonce = best; return ZP_OK; }
    }
}

int32_t zp_merkle_commit(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, uint64_t *d_tree) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "merkle_commit");
    ZP_ARG(ctx, M >= 1 && (M & (M - 1)) == 0, "M must be a power of two");
    ZP_ARG(ctx, W >= 1, "W must be >= 1");
    ZP_ARG(ctx, d_cols && d_tree, "null device pointer");
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    if (M <= ((size_t)1 << 14) && W > 4)
        hipLaunchKernelGGL(merkle_leaves_coop_kernel, dim3((unsigned)((M + 63) / 64)), dim3(768), 0, ctx->stream, (const u64 *)d_cols, M,
                           (size_t)W, M, (size_t)1, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    else if (ctx->mds_is_default)
        hipLaunchKernelGGL(merkle_leaves_kernel<true>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_cols, M, (int)W, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    else
        hipLaunchKernelGGL(merkle_leaves_kernel<false>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_cols, M, (int)W, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return tree_levels(ctx, (u64 *)d_tree, M);
}

int32_t zp_merkle_commit_rows(zp_ctx *ctx, const uint64_t *d_rows, size_t M, size_t len, uint64_t *d_tree) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "merkle_commit_rows");
    ZP_ARG(ctx, M >= 1 && (M & (M - 1)) == 0, "M must be a power of two");
    ZP_ARG(ctx, len >= 1, "len must be >= 1");
    ZP_ARG(ctx, d_rows && d_tree, "null device pointer");
    ZP_TRY(zpi_poseidon_sync_tables(ctx));
    if (M <= ((size_t)1 << 14) && len > 4)
        hipLaunchKernelGGL(merkle_leaves_coop_kernel, dim3((unsigned)((M + 63) / 64)), dim3(768), 0, ctx->stream, (const u64 *)d_rows, M,
                           len, (size_t)1, len, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    else if (ctx->mds_is_default)
        hipLaunchKernelGGL(merkle_leaves_rows_kernel<true>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_rows, M, len, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    else
        hipLaunchKernelGGL(merkle_leaves_rows_kernel<false>, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_rows, M, len, (u64 *)d_tree, ctx->d_rc, ctx->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return tree_levels(ctx, (u64 *)d_tree, M);
}

int32_t zp_merkle_open(zp_ctx *ctx, const uint64_t *d_tree, size_t M, size_t idx, uint64_t *h_path) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, M >= 1 && (M & (M - 1)) == 0, "M must be a power of two");
    ZP_ARG(ctx, idx < M, "leaf index out of range");
    ZP_ARG(ctx, d_tree && h_path, "null pointer");
    const u64 *lvl = (const u64 *)d_tree;
    size_t cnt = M;
    int d = 0;
    while (cnt > 1) {
        ZP_HIP(ctx, hipMemcpyAsync(h_path + 4 * d, lvl + (idx ^ 1) * 4, 4 * sizeof(u64), hipMemcpyDeviceToHost,
                                   ctx->stream));
        lvl += cnt * 4;
        cnt >>= 1;
        idx >>= 1;
        d++;
    }
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

int32_t zp_merkle_commit_host(zp_ctx *ctx, const uint64_t *h_cols, size_t M, int32_t W, uint64_t *h_tree) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, M >= 1 && (M & (M - 1)) == 0, "M must be a power of two");
    ZP_ARG(ctx, W >= 1 && h_cols && h_tree, "bad arguments");
    const size_t bin = (size_t)W * M * sizeof(u64), bt = (2 * M - 1) * 4 * sizeof(u64);
    void *dc = nullptr, *dt = nullptr;
    ZP_TRY(zp_dev_alloc(ctx, bin, &dc));
    int32_t rc = zp_dev_alloc(ctx, bt, &dt);
    if (rc == ZP_OK) rc = zp_h2d(ctx, dc, h_cols, bin);
    if (rc == ZP_OK) rc = zp_merkle_commit(ctx, (const uint64_t *)dc, M, W, (uint64_t *)dt);
    if (rc == ZP_OK) rc = zp_d2h(ctx, h_tree, dt, bt);
    (void)zp_dev_free(ctx, dc);
    (void)zp_dev_free(ctx, dt);
    return rc;
}

}  // extern "C"
