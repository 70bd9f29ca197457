// BN128-hash mode ("final STARK before the Groth16 wrap"): Poseidon over the BN254 scalar field,
// x^5 S-box, 8 full + R_P partial rounds (t = 3: 57, t = 17: 68), and the 16-ary Merkle tree over Goldilocks columns
// packed three to a field element.  The request this serves is GenFinalProof (proto/prover/v1/prover.proto:130-148,
// src/prover/provider.rs:472-503); the reference holds none of it.
// Anchor: with the Grain-LFSR tables of eigen_zeth_amd/poseidon_constants.py the t = 3 instance reproduces the published
// value poseidon([1, 2]) = 0x115cc0f5...189a, and the t = 17 instance -- the width used here -- the published 16-input value
// poseidon([1 .. 16]) = 9989051620...9211877 (tests/test_poseidon_constants.py, tests/test_gpu_bn254_hash.py).
// Lane-per-permutation kernel (commitments of the final STARK): partial rounds in blocks of four (bulk_partial_block, round 6).
//
// Mapping: T lanes per permutation (lane e owns state element e in nine 29-bit Montgomery limbs), floor(64/T)
// permutations per 64-lane workgroup; a full round is  ARK -> S-box (lane-local) -> state to LDS -> each lane one row of the
// dense t x t matrix (T + 3 products deep).
// t = 17, partial rounds: the equivalent SPARSE form (computed by zp_set_poseidon_bn254 from the textbook tables, bit-identical
// results): M = S_i D_i with D_i = diag(1, M^_i) commuting with the one-element S-box and absorbed by the previous round's
// matrix, so a partial round is  s_0 += a_j; s_0 <- s_0^5;  s_0' = m00 s_0 + sum_k v^_k s_k;  s_k' = w_k s_0 + s_k :
// five products deep (x^2 together with the v^_k s_k of the other lanes, x^4, x^5, the column products, one Montgomery
// reduction of the limb-wise row sum) instead of 20.  Per permutation 8 x 20 + 68 x 5 product steps instead of 76 x 20.
#include <hip/hip_runtime.h>

#include <cstring>
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "ctx.hpp"
#include "fr254.hpp"

namespace {

struct P254Table {    // device tables of one instance, Montgomery form, 9 x u32 per element
    int t = 0, rp = 0;
    u32 *d_rc = nullptr;    // [(8 + rp) * t][9]   textbook round constants
    u32 *d_mds = nullptr;   // [t * t][9]   row-major: out[i] = sum_j mds[i][j] * in[j]
    // sparse form of the partial rounds (t = 17 only; nullptr otherwise)
    u32 *d_pre = nullptr;   // [t * t][9]   D_(rp-1) M: the matrix of the last full round before the partial rounds
    u32 *d_a = nullptr;     // [rp][9]      the one constant of partial round j (element 0)
    u32 *d_sp = nullptr;    // [rp][2t-1][9]  per partial round: m00, v^_1 .. v^_(t-1), w_1 .. w_(t-1)
    u32 *d_rcb = nullptr;   // [t][9]       constants of the first full round after the partial rounds (+ the carried rest)
    // the SCALED sparse form (round 5, cooperative kernels: perm_coop): per partial round A'_(j+1), V'_(j,1..t-1), W'_(j,1..t-1); tail: 1 / lambda_rp
    u32 *d_sps = nullptr;   // [rp][2t-1][9]
    u32 *d_spt = nullptr;   // [1][9]
    // the partial rounds in BLOCKS (round 6, lane-per-permutation kernel: bulk_partial_block, NB rounds per block): per block the cross
    // terms C_(J,i) = v^_(j0+J) . w_(j0+i) (i < J), then w_k of the block's rounds for k = 1 .. t-1
    u32 *d_blk = nullptr;   // [rp / NB][NB (NB-1) / 2 + NB (t-1)][9]
};
struct P254Dev { const u32 *rc, *mds, *pre, *a, *sp, *rcb, *sps, *spt; int rp; const u32 *blk; };
// Installed tables: per DEVICE (a single-process multi-GPU host has a ctx per GPU), [0]: t = 3, [1]: t = 17; public constants, shared by
// every ctx of the device.  A published table is never changed or freed: installing the same constants again is a no-op (the ranks of a
// sharded proof all install before they prove), installing different ones publishes a new table and retires the old one, which a kernel
// of another ctx may still be reading (a few hundred KiB per retired table; only tests install twice).
constexpr int P254_MAX_DEV = 64;
const P254Table *g_tables[P254_MAX_DEV][2];
std::vector<u32> g_installed[P254_MAX_DEV][2];      // rc | mds as installed (Montgomery words): what "the same constants" is compared with
std::mutex g_tables_mu;
const P254Table *table_of(const zp_ctx *ctx, int k) {
    if (ctx->device < 0 || ctx->device >= P254_MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lk(g_tables_mu);
    return g_tables[ctx->device][k];
}

__device__ __forceinline__ fr fr_load(const u32 *p) {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}
// sum of up to 18 normalised values (limb-wise in acc) -> the value mod r, normalised: a quotient estimate from the top limb (r's is 0x30644e:
// the estimate is at most one short), one multiply-subtract row, one conditional subtraction -- a third of a Montgomery product, which is what
// reducing such a sum by fr_mul(x, 1) costs
__device__ __forceinline__ fr fr_reduce_small(const u64 *acc) {
    u32 l[9];
    u64 carry = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const u64 v = acc[i] + carry;
        l[i] = i < 8 ? ((u32)v & FR_MASK) : (u32)v;
        carry = v >> FR_B;
    }
    const u32 qe = l[8] / 0x30644fu;
    u32 t[9];
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const long long v = (long long)l[i] - (long long)((u64)qe * fr_p(i)) + c;
        t[i] = i < 8 ? ((u32)v & FR_MASK) : (u32)v;
        c = v >> FR_B;
    }
    return fr_norm_sub(t);                                 // value in [0, 2 r), limbs normalised
}
__device__ __forceinline__ fr sbox5(const fr &x) {
    const fr x2 = fr_sqr(x), x4 = fr_sqr(x2);       // squarings: 45 limb products each instead of 81
    return fr_mul(x4, x);
}

// one full round: ARK (constants c) -> x^5 -> dense matrix m
template <int T>
__device__ __forceinline__ fr full_round(fr s, int q, int e, bool on, u32 (*sh)[T][9], const u32 *c, const u32 *m) {
    if (on) {
        s = sbox5(fr_add(s, fr_load(c + (size_t)e * 9)));
#pragma unroll
        for (int i = 0; i < 9; i++) sh[q][e][i] = s.l[i];
    }
    __syncthreads();
    if (on) {   // row e of the matrix: three products per Montgomery reduction (fr_mul3), T = 3 k + rest
        auto lds = [&](int j) {
            fr v;
#pragma unroll
            for (int i = 0; i < 9; i++) v.l[i] = sh[q][j][i];
            return v;
        };
        const u32 *row = m + (size_t)e * T * 9;
        fr acc = fr_zero();
        int j = 0;
        for (; j + 3 <= T; j += 3)
            acc = fr_add(acc, fr_mul3(fr_load(row + (size_t)j * 9), lds(j), fr_load(row + (size_t)(j + 1) * 9), lds(j + 1),
                                      fr_load(row + (size_t)(j + 2) * 9), lds(j + 2)));
        for (; j < T; j++) acc = fr_add(acc, fr_mul(fr_load(row + (size_t)j * 9), lds(j)));
        s = acc;
    }
    __syncthreads();
    return s;
}

// one permutation of the PPW states a 64-lane workgroup holds; lane = (slot q, element e); sh[q][j] = element j of slot q
template <int T>
__device__ __forceinline__ fr perm_coop(fr s, int q, int e, bool on, u32 (*sh)[T][9], const P254Dev &d) {
    const int rp = d.rp;
    if (d.sp == nullptr) {                      // textbook schedule (t = 3)
        for (int r = 0; r < 8 + rp; r++) {
            if (on) {
                s = fr_add(s, fr_load(d.rc + ((size_t)r * T + e) * 9));
                if (r < 4 || r >= 4 + rp || e == 0) s = sbox5(s);
#pragma unroll
                for (int i = 0; i < 9; i++) sh[q][e][i] = s.l[i];
            }
            __syncthreads();
            if (on) {
                fr acc = fr_zero();
                for (int j = 0; j < T; j++) {
                    fr v;
#pragma unroll
                    for (int i = 0; i < 9; i++) v.l[i] = sh[q][j][i];
                    acc = fr_add(acc, fr_mul(fr_load(d.mds + ((size_t)e * T + j) * 9), v));
                }
                s = acc;
            }
            __syncthreads();
        }
        return s;
    }
    for (int r = 0; r < 4; r++) s = full_round<T>(s, q, e, on, sh, d.rc + (size_t)r * T * 9, r == 3 ? d.pre : d.mds);
    if (d.sps != nullptr) {
        // SCALED sparse form (round 5): lane 0 carries Y_j = lambda_j (s_0 + a_j) with lambda_(j+1) = lambda_j^5 / m00_j, so that
        //     Y_(j+1) = Y_j^5 + sum_k V'_(j,k) s_k + A'_(j+1),      s_k <- s_k + W'_(j,k) Y_j^5      (V' = lambda_(j+1) v^, W' = w / lambda_j^5, A' = lambda a)
        // -- THREE dependent products per round on the chain (Y^2, Y^4, Y^5) where the unscaled form below has five (x^2, x^4, x^5, m00 x^5 and the
        // reduction of the row sum by a product with 1); the other lanes' two products (their update by the previous round's Y^5, then V' s_k)
        // ride in the same instructions, and the row sum is reduced by fr_reduce_small.  Same field values on leaving the loop.
        fr y = s, P = fr_zero();
        if (on && e == 0) y = fr_add(s, fr_load(d.a));                       // Y_0 = s_0 + a_0 (lambda_0 = 1)
        for (int j = 0; j < rp; j++) {
            const u32 *S = d.sps + (size_t)j * (2 * T - 1) * 9;
            const u32 *Sp = d.sps + (size_t)(j ? j - 1 : 0) * (2 * T - 1) * 9;
            if (on) {
                // product 1: lane 0 Y^2; lane k: W'_(j-1,k) Y_(j-1)^5 (P = 0 in round 0), added to s_k
                const fr m1 = fr_mul(e == 0 ? y : fr_load(Sp + (size_t)(T - 1 + e) * 9), e == 0 ? y : P);
                if (e != 0) s = fr_add(s, m1);
                // product 2: lane 0 Y^4; lane k: V'_(j,k) s_k
                fr m2 = fr_mul(e == 0 ? m1 : fr_load(S + (size_t)e * 9), e == 0 ? m1 : s);
                if (e == 0) m2 = fr_mul(m2, y);                              // product 3: Y^5
#pragma unroll
                for (int i = 0; i < 9; i++) sh[q][e][i] = m2.l[i];
            }
            __syncthreads();
            if (on) {
#pragma unroll
                for (int i = 0; i < 9; i++) P.l[i] = sh[q][0][i];
                if (e == 0) {
                    u64 acc[9];
                    const fr ap = fr_load(S);
#pragma unroll
                    for (int i = 0; i < 9; i++) acc[i] = (u64)P.l[i] + ap.l[i];
                    for (int k = 1; k < T; k++) {
#pragma unroll
                        for (int i = 0; i < 9; i++) acc[i] += sh[q][k][i];
                    }
                    y = fr_reduce_small(acc);
                }
            }
            __syncthreads();
        }
        if (on) {                                                            // leave the scaling: s_0 = Y_rp / lambda_rp; s_k takes the last Y^5
            const u32 *Sl = d.sps + (size_t)(rp - 1) * (2 * T - 1) * 9;
            const fr m = fr_mul(e == 0 ? y : fr_load(Sl + (size_t)(T - 1 + e) * 9), e == 0 ? fr_load(d.spt) : P);
            s = e == 0 ? m : fr_add(s, m);
        }
    } else
    for (int j = 0; j < rp; j++) {
        const u32 *sp = d.sp + (size_t)j * (2 * T - 1) * 9;
        if (on) {
            // step A (all lanes, one product): lane 0 squares s_0 + a_j, lane k forms v^_k s_k
            const fr x = e == 0 ? fr_add(s, fr_load(d.a + (size_t)j * 9)) : s;
            const fr y = e == 0 ? x : fr_load(sp + (size_t)e * 9);
            fr p = fr_mul(x, y);
            if (e == 0) {                       // steps B, C: x^4, x^5
                p = fr_mul(p, p);
                p = fr_mul(p, x);
            }
#pragma unroll
            for (int i = 0; i < 9; i++) sh[q][e][i] = p.l[i];
        }
        __syncthreads();
        if (on) {
            fr s0;
#pragma unroll
            for (int i = 0; i < 9; i++) s0.l[i] = sh[q][0][i];
            // step D (all lanes): lane 0: m00 s_0', lane k: w_k s_0'
            const fr c = fr_mul(fr_load(sp + (size_t)(e == 0 ? 0 : T - 1 + e) * 9), s0);
            if (e != 0) {
                s = fr_add(c, s);
            } else {                            // row sum: limb-wise over the T - 1 products, one Montgomery reduction
                u64 acc[9];
#pragma unroll
                for (int i = 0; i < 9; i++) acc[i] = c.l[i];
                for (int k = 1; k < T; k++) {
#pragma unroll
                    for (int i = 0; i < 9; i++) acc[i] += sh[q][k][i];
                }
                fr x;
                u64 carry = 0;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    const u64 v = acc[i] + carry;
                    x.l[i] = i < 8 ? ((u32)v & FR_MASK) : (u32)v;       // value < T r < 2^261
                    carry = v >> FR_B;
                }
                s = fr_mul(x, fr_one());        // x R / R = x mod r (same representation)
            }
        }
        __syncthreads();
    }
    s = full_round<T>(s, q, e, on, sh, d.rcb, d.mds);
    for (int r = 5 + rp; r < 8 + rp; r++) s = full_round<T>(s, q, e, on, sh, d.rc + (size_t)r * T * 9, d.mds);
    return s;
}

// ---- bulk form, t = 17: ONE LANE PER PERMUTATION (launches of >= 2^14 permutations: the Merkle commitments of the final STARK).
// One sponge block of a leaf absorbs up to 56 Goldilocks values in its 16 rate elements: element k holds values base + 3k .. 3k+2 in
// bits 0..191 and, in bits 192..223, 32-bit half number k of values base + 48 .. base + 55 (half 2i = low word of value 48 + i, half
// 2i + 1 = its high word).  Below 2^224 < r.  Rows of at most 48 values pack as three per element.  (oracle/naive.py: pack_leaf_block)
__device__ __forceinline__ void leaf_block_element(const u64 *__restrict__ cols, size_t M, size_t i, int W, int base, int k, u64 *w) {
#pragma unroll
    for (int c = 0; c < 3; c++) w[c] = (base + 3 * k + c < W) ? cols[(size_t)(base + 3 * k + c) * M + i] : 0ULL;
    const int x = base + 48 + (k >> 1);
    w[3] = 0;
    if (x < W) {
        const u64 v = cols[(size_t)x * M + i];
        w[3] = (k & 1) ? (v >> 32) : (v & 0xFFFFFFFFULL);
    }
}

// The cooperative form above gives a wave three permutations (51 of 64 lanes) and makes the whole wave walk the three dependent
// products of the ONE S-box of a partial round: 5 product steps per lane and round for 36 field products per permutation.  Here every
// lane does exactly the products of its own permutation.  The 17-element state lives in LDS ([element][limb][lane]: conflict-free,
// lane-private, no barriers) so that every loop over the state is ROLLED (dynamic LDS addresses; a loop over registers would have to
// be unrolled for static indices: with the state in 153 VGPRs the kernel was 111 k multiply-adds, took 15 minutes to compile and
// spilled).  Rows of a dense matrix and the row of a sparse round are dot products with six terms per Montgomery reduction
// (fr_dotc), tables are wave-uniform (scalar loads).  The 17 rows of a dense matrix need all 17 inputs, so their results wait in
// registers (a switch on the wave-uniform row index gives the rolled loop static register names) and go back to LDS together.
// Same field values as the cooperative form, hence the same canonical words (both are checked against oracle/bn254_hash.c).
#define P254_BULK_LDS (17 * 9 * 64)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {      // expanded at compile time: static register indices
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// the LDS-resident state is named by address-space-3 pointers: with generic ones the round functions below met a code generator
// fault (an LDS -> flat cast with its null check, "V_CMP_NE_U32 0, $src_shared_base: operand has incorrect register class")
typedef __attribute__((address_space(3))) u32 lu32;
__device__ __forceinline__ void lds_put(lu32 *st, int e, int lane, const fr &x) {
#pragma unroll
    for (int i = 0; i < 9; i++) st[(e * 9 + i) * 64 + lane] = x.l[i];
}
__device__ __forceinline__ fr lds_get(const lu32 *st, int e, int lane) {
    fr x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.l[i] = st[(e * 9 + i) * 64 + lane];
    return x;
}
// sum_j row[j] * state[j] over the LDS-resident state, 6 + 6 + 5 terms; with HAS_FIRST `first` stands in for element 0 (the S-box
// output of a partial round, not written back yet)
template <bool HAS_FIRST>
__device__ __forceinline__ fr dot17_lds(const lu32 *st, int lane, const u32 *__restrict__ row, const fr &first) {
    fr acc;
#pragma unroll 1
    for (int g = 0; g < 2; g++) {
        fr x[6];
#pragma unroll
        for (int jj = 0; jj < 6; jj++) x[jj] = lds_get(st, g * 6 + jj, lane);
        if (HAS_FIRST && g == 0) x[0] = first;
        const fr d = fr_dotc<6>(x, row + g * 6 * 9);
        acc = g == 0 ? d : fr_add(acc, d);
    }
    fr x[5];
#pragma unroll
    for (int jj = 0; jj < 5; jj++) x[jj] = lds_get(st, 12 + jj, lane);
    return fr_add(acc, fr_dotc<5>(x, row + 12 * 9));
}
// One full round on the LDS-resident state: ARK -> x^5 -> dense matrix (the sparse form moves constants / matrices around them)
// ONLY0: the caller reads element 0 alone afterwards (the LAST round of a sponge block or a tree node: the digest / next capacity) --
// one row of the dense matrix instead of seventeen; elements 1..16 are left stale
template <bool ONLY0 = false>
__device__ __forceinline__ void bulk_full_round(lu32 *st, int lane, const u32 *c, const u32 *m) {
    const fr zero = fr_zero();
#pragma unroll 1
    for (int e = 0; e < 17; e++) lds_put(st, e, lane, sbox5(fr_add(lds_get(st, e, lane), fr_load(c + e * 9))));
    if constexpr (ONLY0) {
        lds_put(st, 0, lane, dot17_lds<false>(st, lane, m, zero));
        return;
    }
    fr o[17];
#pragma unroll 1
    for (int e = 0; e < 17; e++) {
        const fr v = dot17_lds<false>(st, lane, m + (size_t)e * 17 * 9, zero);
        switch (e) {                // wave-uniform: static register names for the rolled loop's results
#define P254_CASE(K) case K: o[K] = v; break;
            P254_CASE(0) P254_CASE(1) P254_CASE(2) P254_CASE(3) P254_CASE(4) P254_CASE(5) P254_CASE(6) P254_CASE(7) P254_CASE(8)
            P254_CASE(9) P254_CASE(10) P254_CASE(11) P254_CASE(12) P254_CASE(13) P254_CASE(14) P254_CASE(15)
            default: o[16] = v; break;
#undef P254_CASE
        }
    }
    static_for<0, 17>([&](auto E) { lds_put(st, decltype(E)::value, lane, o[decltype(E)::value]); });
}
// Partial rounds j0 .. j0 + NB - 1 in one block (round 6).  Round by round, every round updates the 16 other elements, s_k += w_k sg --
// sixteen single products, each with its own Montgomery reduction and a canonical addition: two thirds of a partial round's instructions.
// The updates are linear in the S-box outputs sg_i, so inside a block the elements stay at their block-start values S_k and round J reads
//     s_0 <- m00 sg_J + sum_k v^_k S_k + sum_(i < J) C_(J,i) sg_i,          C_(J,i) = v^_(j0+J) . w_(j0+i)   (host, once per table)
// and after the last round  S_k += sum_i w_k^(j0+i) sg_i : ONE NB-term dot product (one reduction) per element instead of NB single
// products.  The same field values (every digest identical to the round-by-round form).  NB = 4: the S-box outputs of a block stay in
// 36 registers (with five or six of them live across both loops the register allocator spilled to scratch: slower than round by round;
// three: 34.0 M permutations/s, four: 34.5 M, round by round: 29.3 M -- profiles/r6_p254_block_ab.txt).
#define P254_NB 4
#define P254_BLK_ENTRIES (P254_NB * (P254_NB - 1) / 2 + P254_NB * 16)
__device__ __forceinline__ void bulk_partial_block(lu32 *st, int lane, const P254Dev &d, int b) {
    const int j0 = P254_NB * b;
    const u32 *blk = d.blk + (size_t)b * P254_BLK_ENTRIES * 9;
    fr sg[P254_NB];
    fr s0 = lds_get(st, 0, lane);
    static_for<0, P254_NB>([&](auto Jc) {
        constexpr int J = decltype(Jc)::value;
        const u32 *sp = d.sp + (size_t)(j0 + J) * 33 * 9;
        sg[J] = sbox5(fr_add(s0, fr_load(d.a + (size_t)(j0 + J) * 9)));
        s0 = dot17_lds<true>(st, lane, sp, sg[J]);
        if constexpr (J > 0) s0 = fr_add(s0, fr_dotc<J>(sg, blk + (size_t)(J * (J - 1) / 2) * 9));
    });
#pragma unroll 1
    for (int k = 1; k < 17; k++)
        lds_put(st, k, lane, fr_add(fr_dotc<P254_NB>(sg, blk + (size_t)(P254_NB * (P254_NB - 1) / 2 + (k - 1) * P254_NB) * 9), lds_get(st, k, lane)));
    lds_put(st, 0, lane, s0);
}
template <bool ONLY0 = false>
__device__ __forceinline__ void bulk_perm17(lu32 *st, int lane, const P254Dev &d) {
    const int rp = d.rp;
#pragma unroll 1
    for (int r = 0; r < 4; r++) bulk_full_round(st, lane, d.rc + (size_t)r * 17 * 9, r == 3 ? d.pre : d.mds);
    const int nblk = d.blk ? rp / P254_NB : 0;
#pragma unroll 1
    for (int b = 0; b < nblk; b++) bulk_partial_block(st, lane, d, b);
#pragma unroll 1
    for (int j = P254_NB * nblk; j < rp; j++) {  // partial round j: s_0 <- (s_0 + a_j)^5; s_0' = m00 s_0 + sum_k v^_k s_k; s_k' = w_k s_0 + s_k
        const u32 *sp = d.sp + (size_t)j * 33 * 9;
        const fr sg = sbox5(fr_add(lds_get(st, 0, lane), fr_load(d.a + (size_t)j * 9)));
        const fr n0 = dot17_lds<true>(st, lane, sp, sg);
#pragma unroll 1
        for (int k = 1; k < 17; k++) lds_put(st, k, lane, fr_add(fr_dotc<1>(&sg, sp + (16 + k) * 9), lds_get(st, k, lane)));
        lds_put(st, 0, lane, n0);
    }
#pragma unroll 1
    for (int r = 4 + rp; r < 7 + rp; r++) bulk_full_round(st, lane, r == 4 + rp ? d.rcb : d.rc + (size_t)r * 17 * 9, d.mds);
    bulk_full_round<ONLY0>(st, lane, d.rc + (size_t)(7 + rp) * 17 * 9, d.mds);
}

// mode 0: states u64[count][17][4] permuted in place;  mode 1: leaves of the 16-ary tree (lane = row, column reads are coalesced
// 512-byte runs);  mode 2: one tree level (16 children per node)
template <int MODE>
__global__ void __launch_bounds__(64) p254_bulk_kernel(const u64 *__restrict__ in, size_t n_in, int W, u64 *__restrict__ out, size_t count, P254Dev d) {
    __shared__ u32 st_mem[P254_BULK_LDS];
    lu32 *const st = (lu32 *)st_mem;
    const int lane = threadIdx.x;
    const size_t i = (size_t)blockIdx.x * 64 + lane;
    const bool on = i < count;
    const size_t ii = on ? i : 0;
    if constexpr (MODE == 0) {
#pragma unroll 1
        for (int e = 0; e < 17; e++) {
            u64 w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = out[(ii * 17 + e) * 4 + k];
            lds_put(st, e, lane, fr_to_mont(fr_from_u64(w)));
        }
        bulk_perm17(st, lane, d);
#pragma unroll 1
        for (int e = 0; e < 17; e++) {
            u64 w[4];
            fr_to_u64(fr_from_mont(lds_get(st, e, lane)), w);
            if (on) {
#pragma unroll
                for (int k = 0; k < 4; k++) out[(i * 17 + e) * 4 + k] = w[k];
            }
        }
    } else {
        lds_put(st, 0, lane, fr_zero());
        if constexpr (MODE == 1) {          // in = cols u64[W][M], M = count = n_in; sponge over blocks of 56 values in 16 elements
#pragma unroll 1
            for (int base = 0; base < W || base == 0; base += 56) {
#pragma unroll 1
                for (int e = 1; e < 17; e++) {
                    u64 w[4];
                    leaf_block_element(in, (size_t)n_in, ii, W, base, e - 1, w);
                    lds_put(st, e, lane, fr_to_mont(fr_from_u64(w)));
                }
                bulk_perm17<true>(st, lane, d);   // the digest (element 0) stays as the capacity of the next block: the only element read
            }
        } else {                            // in = previous level u64[n_in][4]
#pragma unroll 1
            for (int e = 1; e < 17; e++) {
                const size_t c = ii * 16 + (e - 1);
                u64 w[4] = {0, 0, 0, 0};
                if (c < n_in) {
#pragma unroll
                    for (int k = 0; k < 4; k++) w[k] = in[c * 4 + k];
                }
                lds_put(st, e, lane, fr_to_mont(fr_from_u64(w)));
            }
            bulk_perm17<true>(st, lane, d);
        }
        u64 w[4];
        fr_to_u64(fr_from_mont(lds_get(st, 0, lane)), w);
        if (on) {
#pragma unroll
            for (int k = 0; k < 4; k++) out[i * 4 + k] = w[k];
        }
    }
}

// states: u64[count][T][4] standard form, permuted in place
template <int T>
__global__ void __launch_bounds__(64) poseidon254_perm_kernel(u64 *states, size_t count, P254Dev d) {
    constexpr int PPW = 64 / T;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t idx = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && idx < count;
    fr s = fr_zero();
    if (on) {
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = states[(idx * T + e) * 4 + k];
        s = fr_to_mont(fr_from_u64(w));
    }
    s = perm_coop<T>(s, q, e, on, sh, d);
    if (on) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) states[(idx * T + e) * 4 + k] = w[k];
    }
}

// R1CS witness of the width-17 gadget (eigen_zeth_amd/service/r1cs.py: poseidon_template; csrc/r1cs.hip: zp_r1cs_eval_device): an instance of the
// gadget is a permutation whose every S-box leaves three internal wires (x^2, x^4, x^5 of its input x) and three constraint rows
// (x, x, x^2), (x^2, x^2, x^4), (x^4, x, x^5) -- the S-boxes numbered in the TEXTBOOK order (rounds, then elements) --, and one last wire / row for
// element 0 of the output (out, 1, out).  So the witness and A w, B w, C w of an instance are the intermediate values of the permutation: 17 lanes
// per instance walk the textbook schedule (not the sparse form: the template's linear combinations are the textbook's) and write them out.
// inst: per instance 17 input wires, the first internal wire, the first row.  flags[1] <- the smallest row of an instance that reads an unset wire.
__global__ void __launch_bounds__(64) r1cs_poseidon17_kernel(const u64 *__restrict__ inst, size_t i0, size_t count, u64 *__restrict__ w, unsigned char *__restrict__ set,
                                                             u64 *__restrict__ a_ev, u64 *__restrict__ b_ev, u64 *__restrict__ c_ev, unsigned long long *flags, P254Dev d) {
    constexpr int T = 17, PPW = 3;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t local = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && local < count;
    const u64 *in = inst + (i0 + (on ? local : 0)) * (T + 2);
    const u64 base = in[T], row0 = in[T + 1];
    fr s = fr_zero();
    if (on) {
        const u64 wire = in[e];
        if (!set[wire]) atomicMin(&flags[1], (unsigned long long)row0);
        u64 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = w[wire * 4 + k];
        s = fr_to_mont(fr_from_u64(v));
    }
    auto put = [](u64 *dst, u64 index, const fr &mont) {
        u64 v[4];
        fr_to_u64(fr_from_mont(mont), v);
#pragma unroll
        for (int k = 0; k < 4; k++) dst[index * 4 + k] = v[k];
    };
    const int rp = d.rp;
    for (int r = 0; r < 8 + rp; r++) {
        const bool full = r < 4 || r >= 4 + rp;
        if (on) {
            s = fr_add(s, fr_load(d.rc + ((size_t)r * T + e) * 9));
            if (full || e == 0) {
                const u64 k = r < 4 ? (u64)r * T + e : !full ? (u64)(4 * T + (r - 4)) : (u64)(4 * T + rp + (r - 4 - rp) * T + e);
                const fr x = s, x2 = fr_mul(x, x), x4 = fr_mul(x2, x2), x5 = fr_mul(x4, x);
                put(w, base + 3 * k, x2); put(w, base + 3 * k + 1, x4); put(w, base + 3 * k + 2, x5);
                set[base + 3 * k] = 1; set[base + 3 * k + 1] = 1; set[base + 3 * k + 2] = 1;
                const u64 row = row0 + 3 * k;
                put(a_ev, row, x); put(b_ev, row, x); put(c_ev, row, x2);
                put(a_ev, row + 1, x2); put(b_ev, row + 1, x2); put(c_ev, row + 1, x4);
                put(a_ev, row + 2, x4); put(b_ev, row + 2, x); put(c_ev, row + 2, x5);
                s = x5;
            }
#pragma unroll
            for (int i = 0; i < 9; i++) sh[q][e][i] = s.l[i];
        }
        __syncthreads();
        if (on) {
            auto lds = [&](int j) {
                fr v;
#pragma unroll
                for (int i = 0; i < 9; i++) v.l[i] = sh[q][j][i];
                return v;
            };
            const u32 *row = d.mds + (size_t)e * T * 9;
            fr acc = fr_zero();
            int j = 0;
            for (; j + 3 <= T; j += 3)
                acc = fr_add(acc, fr_mul3(fr_load(row + (size_t)j * 9), lds(j), fr_load(row + (size_t)(j + 1) * 9), lds(j + 1), fr_load(row + (size_t)(j + 2) * 9), lds(j + 2)));
            for (; j < T; j++) acc = fr_add(acc, fr_mul(fr_load(row + (size_t)j * 9), lds(j)));
            s = acc;
        }
        __syncthreads();
    }
    if (on && e == 0) {
        const u64 k = (u64)(8 * T + rp);                  // S-boxes in all
        put(w, base + 3 * k, s);
        set[base + 3 * k] = 1;
        fr one = fr_zero();
        one.l[0] = 1;
        put(a_ev, row0 + 3 * k, s); put(c_ev, row0 + 3 * k, s);
        u64 *b = b_ev + (row0 + 3 * k) * 4;
        b[0] = 1; b[1] = 0; b[2] = 0; b[3] = 0;
    }
}

// The width-17 transcript sponge as ONE launch (17 lanes cooperate on the one state): buf = [17 state elements][nblocks x 16 block
// elements][(1 + extra) x 16 rate elements out], 4 words each, standard form.  For every block the rate (elements 1..16) is
// overwritten with the block and the state permuted (no block: one permutation), then `extra` more permutations; the rate after the
// absorption and after every extra permutation is written out.  A transcript step of k permutations costs one host round trip
// instead of k launches with a copy between them.
// caps (may be null): the capacity element after EVERY permutation, in order -- what a circuit that re-hashes the transcript with all its
// gadgets side by side (service/wrap_circuit.py: every gadget's capacity input a caller-set wire) has to be told
__global__ void __launch_bounds__(64) poseidon254_sponge_kernel(u64 *buf, int nblocks, int extra, u64 *caps, P254Dev d) {
    constexpr int T = 17;
    __shared__ u32 sh[3][T][9];
    const int e = threadIdx.x;
    const bool on = e < T;
    u64 *rates = buf + (size_t)(17 + (size_t)nblocks * 16) * 4;
    fr s = fr_zero();
    if (on) {
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = buf[(size_t)e * 4 + k];
        s = fr_to_mont(fr_from_u64(w));
    }
    const int absorb = nblocks > 0 ? nblocks : 1;
    for (int b = 0; b < absorb + extra; b++) {
        if (b < nblocks && on && e >= 1) {
            u64 w[4];
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = buf[(size_t)(17 + (size_t)b * 16 + (e - 1)) * 4 + k];
            s = fr_to_mont(fr_from_u64(w));
        }
        s = perm_coop<T>(s, 0, on ? e : 0, on, sh, d);
        if (caps && e == 0) {
            u64 w[4];
            fr_to_u64(fr_from_mont(s), w);
#pragma unroll
            for (int k = 0; k < 4; k++) caps[(size_t)b * 4 + k] = w[k];
        }
        if (b >= absorb - 1 && on && e >= 1) {
            u64 w[4];
            fr_to_u64(fr_from_mont(s), w);
#pragma unroll
            for (int k = 0; k < 4; k++) rates[((size_t)(b - (absorb - 1)) * 16 + (e - 1)) * 4 + k] = w[k];
        }
    }
    if (on) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) buf[(size_t)e * 4 + k] = w[k];
    }
}

// 16-ary Merkle tree, t = 17: state = [capacity, 16 inputs]; digest = state[0] after the permutation.
// leaves: row i of the Goldilocks matrix cols[W][M], three values per field element (a + b 2^64 + c 2^128), sponge over
// blocks of 16 elements with the digest as the next capacity.  nodes: 16 child digests (missing children = 0).
__global__ void __launch_bounds__(64) merkle16_leaves_kernel(const u64 *__restrict__ cols, size_t M, int W, u64 *__restrict__ tree, P254Dev d) {
    constexpr int T = 17, PPW = 3;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t i = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && i < M;
    fr s = fr_zero();                                  // capacity 0
    for (int base = 0; base < W || base == 0; base += 56) {
        if (on && e >= 1) {
            u64 w[4];
            leaf_block_element(cols, M, i, W, base, e - 1, w);
            s = fr_to_mont(fr_from_u64(w));
        }
        s = perm_coop<T>(s, q, e, on, sh, d);
        // the digest (element 0) is the capacity of the next block: lane e = 0 already holds it
    }
    if (on && e == 0) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) tree[i * 4 + k] = w[k];
    }
}

__global__ void __launch_bounds__(64) merkle16_level_kernel(const u64 *__restrict__ prev, size_t nprev, u64 *__restrict__ next, size_t nnext, P254Dev d) {
    constexpr int T = 17, PPW = 3;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t i = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && i < nnext;
    fr s = fr_zero();
    if (on && e >= 1) {
        const size_t c = i * 16 + (e - 1);
        u64 w[4] = {0, 0, 0, 0};
        if (c < nprev) {
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = prev[c * 4 + k];
        }
        s = fr_to_mont(fr_from_u64(w));
    }
    s = perm_coop<T>(s, q, e, on, sh, d);
    if (on && e == 0) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) next[i * 4 + k] = w[k];
    }
}

// openings of nq leaves in one launch: out u64[nq][levels][16][4], per level the 16 digests of the group on the path (children
// beyond the level read as zero)
__global__ void __launch_bounds__(256) merkle16_paths_kernel(const u64 *__restrict__ tree, u64 M, int levels, const u64 *__restrict__ idx, int nq,
                                                            u64 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= (u64)nq * levels * 64) return;
    const int word = (int)(i & 3), child = (int)((i >> 2) & 15);
    const u64 ql = i >> 6;
    const int lvl = (int)(ql % levels);
    const u64 q = ql / levels;
    u64 n = M, off = 0, pos = idx[q];
    for (int l = 0; l < lvl; l++) { off += n; n = (n + 15) / 16; pos /= 16; }
    const u64 c = (pos / 16) * 16 + child;
    out[i] = c < n ? tree[(off + c) * 4 + word] : 0ULL;
}

int32_t table_for(zp_ctx *ctx, int t, const P254Table **out) {
    ZP_ARG(ctx, t == 3 || t == 17, "Poseidon-BN254 width must be 3 or 17");
    const P254Table *tb = table_of(ctx, t == 3 ? 0 : 1);
    ZP_ARG(ctx, tb != nullptr && tb->d_rc != nullptr, "Poseidon-BN254 tables not installed on this device (zp_set_poseidon_bn254)");
    *out = tb;
    return ZP_OK;
}
// permutations from which the lane-per-permutation kernel takes over (zp_set_tuning "p254_bulk_log"; 0 = 2^14; 31 = never): one wave
// on every SIMD of the chip takes 2^16 permutations, below ~2^14 the cooperative form's shorter dependency chains win
int bulk_threshold(zp_ctx *ctx) {
    const int lg = ctx->tune_p254_bulk_log > 0 ? ctx->tune_p254_bulk_log : 14;
    return lg >= 31 ? 0x7FFFFFFF : (1 << lg);
}
P254Dev dev_of(const zp_ctx *ctx, const P254Table *tb) {       // (zp_set_tuning "p254_scaled" = 2: the cooperative kernels on the unscaled sparse form, for the A/B)
    const bool scaled = ctx->tune_p254_scaled != 2 && tb->d_sps && tb->d_spt;
    return P254Dev{tb->d_rc, tb->d_mds, tb->d_pre, tb->d_a, tb->d_sp, tb->d_rcb, scaled ? tb->d_sps : nullptr, scaled ? tb->d_spt : nullptr, tb->rp,
                   ctx->tune_p254_block == 2 ? nullptr : tb->d_blk};      // (zp_set_tuning "p254_block" = 2: round by round, for the A/B)
}

// ---- host: the sparse form of the partial rounds (column-vector convention, values in Montgomery form) ----
// round j (0 <= j < rp) of the textbook is  x <- M sigma(x + c_j),  sigma = x^5 on element 0.  Backwards from the last round,
// M_0 = M:  M_i = S_i D_i,  D_i = diag(1, M^_i)  (M^_i = M_i without row/column 0),  S_i = M_i D_i^-1 = [[m00, v M^_i^-1], [w, I]].
// D_i commutes with sigma and moves into the previous round: M_(i+1) = D_i M, c_j <- D_i c_j  (j = rp - 1 - i); what is left
// after round 0 is the matrix D_(rp-1) M of the full round before.  Forwards, the constants of elements 1.. pass sigma
// unchanged and are carried through S (carry_0' = v^ . rest, carry_k' = rest_k) into the next round's constants; the last
// carry lands in the constants of the first full round after the partial rounds.
typedef std::vector<fr> FrVec;
fr h_inv_fr(fr a) {
    const u64 e[4] = {0x43e1f593f0000001ULL - 2, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    fr r = fr_one();
    for (int k = 3; k >= 0; k--)
        for (int b = 63; b >= 0; b--) {
            r = fr_mul(r, r);
            if ((e[k] >> b) & 1) r = fr_mul(r, a);
        }
    return r;
}
bool fr_is_zero(const fr &a) {
    u32 o = 0;
    for (int i = 0; i < 9; i++) o |= a.l[i];
    return o == 0;
}
// inverse of an n x n matrix (row-major) by Gauss-Jordan; false if singular
bool mat_inv(const FrVec &a_in, int n, FrVec &out) {
    FrVec a = a_in;
    out.assign((size_t)n * n, fr_zero());
    for (int i = 0; i < n; i++) out[(size_t)i * n + i] = fr_one();
    for (int c = 0; c < n; c++) {
        int piv = -1;
        for (int r = c; r < n; r++)
            if (!fr_is_zero(a[(size_t)r * n + c])) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != c)
            for (int k = 0; k < n; k++) { std::swap(a[(size_t)piv * n + k], a[(size_t)c * n + k]); std::swap(out[(size_t)piv * n + k], out[(size_t)c * n + k]); }
        const fr pi = h_inv_fr(a[(size_t)c * n + c]);
        for (int k = 0; k < n; k++) { a[(size_t)c * n + k] = fr_mul(a[(size_t)c * n + k], pi); out[(size_t)c * n + k] = fr_mul(out[(size_t)c * n + k], pi); }
        for (int r = 0; r < n; r++) {
            if (r == c || fr_is_zero(a[(size_t)r * n + c])) continue;
            const fr f = a[(size_t)r * n + c];
            for (int k = 0; k < n; k++) {
                a[(size_t)r * n + k] = fr_sub(a[(size_t)r * n + k], fr_mul(f, a[(size_t)c * n + k]));
                out[(size_t)r * n + k] = fr_sub(out[(size_t)r * n + k], fr_mul(f, out[(size_t)c * n + k]));
            }
        }
    }
    return true;
}
// rc: (8 + rp) t constants round-major, M: t x t.  Outputs: pre (t x t), a (rp), sp (rp x (2t-1)), rcb (t).
bool sparse_form(const FrVec &rc, const FrVec &M, int t, int rp, FrVec &pre, FrVec &a, FrVec &sp, FrVec &rcb) {
    const int n = t - 1;
    FrVec Mi = M, cD((size_t)rp * t);
    sp.assign((size_t)rp * (2 * t - 1), fr_zero());
    for (int i = 0; i < rp; i++) {
        const int j = rp - 1 - i;
        FrVec Mh((size_t)n * n), Mhinv;
        for (int r = 0; r < n; r++)
            for (int c = 0; c < n; c++) Mh[(size_t)r * n + c] = Mi[(size_t)(r + 1) * t + (c + 1)];
        if (!mat_inv(Mh, n, Mhinv)) return false;
        fr *S = &sp[(size_t)j * (2 * t - 1)];
        S[0] = Mi[0];                                                    // m00
        for (int k = 0; k < n; k++) {                                    // v^ = v M^^-1
            fr acc = fr_zero();
            for (int l = 0; l < n; l++) acc = fr_add(acc, fr_mul(Mi[(size_t)(l + 1)], Mhinv[(size_t)l * n + k]));
            S[1 + k] = acc;
            S[t + k] = Mi[(size_t)(k + 1) * t];                          // w_k
        }
        // c_j <- D_i c_j
        const fr *c = &rc[(size_t)(4 + j) * t];
        cD[(size_t)j * t] = c[0];
        for (int r = 0; r < n; r++) {
            fr acc = fr_zero();
            for (int l = 0; l < n; l++) acc = fr_add(acc, fr_mul(Mh[(size_t)r * n + l], c[l + 1]));
            cD[(size_t)j * t + r + 1] = acc;
        }
        // M_(i+1) = D_i M: row 0 of M, rows 1.. = M^ M[1:, :]
        FrVec nx((size_t)t * t);
        for (int c2 = 0; c2 < t; c2++) nx[c2] = M[c2];
        for (int r = 0; r < n; r++)
            for (int c2 = 0; c2 < t; c2++) {
                fr acc = fr_zero();
                for (int l = 0; l < n; l++) acc = fr_add(acc, fr_mul(Mh[(size_t)r * n + l], M[(size_t)(l + 1) * t + c2]));
                nx[(size_t)(r + 1) * t + c2] = acc;
            }
        Mi = nx;
    }
    pre = Mi;
    a.assign(rp, fr_zero());
    FrVec carry(t, fr_zero());
    for (int j = 0; j < rp; j++) {
        const fr *S = &sp[(size_t)j * (2 * t - 1)];
        FrVec tv(t);
        for (int k = 0; k < t; k++) tv[k] = fr_add(cD[(size_t)j * t + k], carry[k]);
        a[j] = tv[0];
        fr acc = fr_zero();
        for (int k = 1; k < t; k++) { acc = fr_add(acc, fr_mul(S[k], tv[k])); carry[k] = tv[k]; }
        carry[0] = acc;
    }
    rcb.resize(t);
    for (int k = 0; k < t; k++) rcb[k] = fr_add(rc[(size_t)(4 + rp) * t + k], carry[k]);
    return true;
}
// the scaled sparse form of the partial rounds (perm_coop): from a (rp) and sp (rp x (2t-1): m00, v^_1.., w_1..) the per-round rows
// [A'_(j+1), V'_(j,k) = lambda_(j+1) v^_(j,k), W'_(j,k) = w_(j,k) / lambda_j^5] and the tail 1 / lambda_rp, lambda_0 = 1, lambda_(j+1) = lambda_j^5 / m00_j,
// A'_j = lambda_j a_j, A'_rp = 0.  false when some m00 is zero.
bool scaled_sparse_form(const FrVec &a, const FrVec &sp, int t, int rp, FrVec &sps, FrVec &spt) {
    sps.assign((size_t)rp * (2 * t - 1), fr_zero());
    fr lam = fr_one();
    for (int j = 0; j < rp; j++) {
        const fr *S = &sp[(size_t)j * (2 * t - 1)];
        if (fr_is_zero(S[0])) return false;
        const fr l2 = fr_mul(lam, lam), l5 = fr_mul(fr_mul(l2, l2), lam);
        const fr next = fr_mul(l5, h_inv_fr(S[0])), l5inv = h_inv_fr(l5);
        fr *O = &sps[(size_t)j * (2 * t - 1)];
        O[0] = j + 1 < rp ? fr_mul(next, a[j + 1]) : fr_zero();
        for (int k = 1; k < t; k++) {
            O[k] = fr_mul(next, S[k]);
            O[t - 1 + k] = fr_mul(S[t - 1 + k], l5inv);
        }
        lam = next;
    }
    spt.assign(1, h_inv_fr(lam));
    return true;
}
int32_t upload_fr(zp_ctx *ctx, const FrVec &v, u32 **d) {
    std::vector<u32> flat(v.size() * 9);
    for (size_t i = 0; i < v.size(); i++) memcpy(&flat[i * 9], v[i].l, 36);
    ZP_HIP(ctx, hipMalloc((void **)d, flat.size() * 4));
    ZP_HIP(ctx, hipMemcpy(*d, flat.data(), flat.size() * 4, hipMemcpyHostToDevice));
    return ZP_OK;
}

}  // namespace

extern "C" {

int32_t zp_set_poseidon_bn254(zp_ctx *ctx, int32_t t, int32_t rp, const uint64_t *h_rc, const uint64_t *h_mds) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, (t == 3 || t == 17) && rp >= 1 && rp <= 128 && h_rc && h_mds, "bad arguments (t must be 3 or 17)");
    const size_t nrc = (size_t)(8 + rp) * t, nm = (size_t)t * t;
    std::vector<u32> rc(nrc * 9), md(nm * 9);
    for (size_t i = 0; i < nrc + nm; i++) {
        const uint64_t *src = i < nrc ? h_rc + 4 * i : h_mds + 4 * (i - nrc);
        u64 w[4] = {src[0], src[1], src[2], src[3]};
        ZP_ARG(ctx, fr_is_canonical_u64(w), "constant not reduced mod r");
        const fr m = fr_to_mont(fr_from_u64(w));
        memcpy((i < nrc ? rc.data() + 9 * i : md.data() + 9 * (i - nrc)), m.l, 36);
    }
    ZP_ARG(ctx, ctx->device >= 0 && ctx->device < P254_MAX_DEV, "device ordinal out of range");
    std::vector<u32> key(rc);
    key.insert(key.end(), md.begin(), md.end());
    key.push_back((u32)rp);
    std::lock_guard<std::mutex> lk(g_tables_mu);           // installs are serialised; readers take the published pointer (table_of)
    const int slot = t == 3 ? 0 : 1;
    if (g_tables[ctx->device][slot] && g_installed[ctx->device][slot] == key) return ZP_OK;      // the same constants: nothing to do
    std::unique_ptr<P254Table> own(new P254Table());       // published below; dropped (its device buffers with it: a failed install is rare) on an error return
    P254Table *tb = own.get();
    if (t == 17) {                                  // sparse form of the partial rounds
        FrVec frc(nrc), fm(nm), pre, a, sp, rcb;
        for (size_t i = 0; i < nrc; i++) memcpy(frc[i].l, &rc[i * 9], 36);
        for (size_t i = 0; i < nm; i++) memcpy(fm[i].l, &md[i * 9], 36);
        ZP_ARG(ctx, sparse_form(frc, fm, t, rp, pre, a, sp, rcb), "matrix has a singular minor: no sparse form (not an MDS matrix)");
        ZP_TRY(upload_fr(ctx, pre, &tb->d_pre));
        ZP_TRY(upload_fr(ctx, a, &tb->d_a));
        ZP_TRY(upload_fr(ctx, sp, &tb->d_sp));
        ZP_TRY(upload_fr(ctx, rcb, &tb->d_rcb));
        if (rp >= P254_NB) {                        // the blocks of partial rounds (bulk_partial_block)
            FrVec blk((size_t)(rp / P254_NB) * P254_BLK_ENTRIES, fr_zero());
            for (int b = 0; b < rp / P254_NB; b++) {
                fr *o = blk.data() + (size_t)b * P254_BLK_ENTRIES;
                for (int J = 1; J < P254_NB; J++)
                    for (int i = 0; i < J; i++) {
                        const fr *v = sp.data() + (size_t)(P254_NB * b + J) * 33 + 1, *w = sp.data() + (size_t)(P254_NB * b + i) * 33 + 17;
                        fr acc = fr_zero();
                        for (int k = 0; k < 16; k++) acc = fr_add(acc, fr_mul(v[k], w[k]));
                        o[J * (J - 1) / 2 + i] = acc;
                    }
                for (int k = 0; k < 16; k++)
                    for (int i = 0; i < P254_NB; i++) o[P254_NB * (P254_NB - 1) / 2 + k * P254_NB + i] = sp[(size_t)(P254_NB * b + i) * 33 + 17 + k];
            }
            ZP_TRY(upload_fr(ctx, blk, &tb->d_blk));
        }
        FrVec sps, spt;
        if (scaled_sparse_form(a, sp, t, rp, sps, spt)) {        // (a zero m00 somewhere: the cooperative kernels keep the unscaled form)
            ZP_TRY(upload_fr(ctx, sps, &tb->d_sps));
            ZP_TRY(upload_fr(ctx, spt, &tb->d_spt));
        }
    }
    ZP_HIP(ctx, hipMalloc((void **)&tb->d_rc, rc.size() * 4));
    ZP_HIP(ctx, hipMalloc((void **)&tb->d_mds, md.size() * 4));
    ZP_HIP(ctx, hipMemcpy(tb->d_rc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice));
    ZP_HIP(ctx, hipMemcpy(tb->d_mds, md.data(), md.size() * 4, hipMemcpyHostToDevice));
    tb->t = t;
    tb->rp = rp;
    g_tables[ctx->device][slot] = own.release();                   // (the table it replaces stays allocated: see g_tables)
    g_installed[ctx->device][slot] = std::move(key);
    return ZP_OK;
}

int32_t zp_poseidon_bn254_perm(zp_ctx *ctx, uint64_t *d_states, size_t count, int32_t t) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_bn254_perm");
    const P254Table *tb;
    ZP_TRY(table_for(ctx, t, &tb));
    if (count == 0) return ZP_OK;
    ZP_ARG(ctx, d_states != nullptr, "null device pointer");
    if (t == 3)
        hipLaunchKernelGGL(poseidon254_perm_kernel<3>, dim3((unsigned)((count + 20) / 21)), dim3(64), 0, ctx->stream, (u64 *)d_states, count,
                           dev_of(ctx, tb));
    else if (count >= (size_t)bulk_threshold(ctx))     // one lane per permutation once there are enough of them to fill the chip
        hipLaunchKernelGGL(p254_bulk_kernel<0>, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, (const u64 *)nullptr, (size_t)0, 0,
                           (u64 *)d_states, count, dev_of(ctx, tb));
    else
        hipLaunchKernelGGL(poseidon254_perm_kernel<17>, dim3((unsigned)((count + 2) / 3)), dim3(64), 0, ctx->stream, (u64 *)d_states, count,
                           dev_of(ctx, tb));
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

// The width-17 sponge of the BN128-mode transcript (element 0 = capacity, 1..16 = rate), one host call per transcript step:
// for each of the nblocks blocks of 16 elements the rate is overwritten with the block and the state permuted (nblocks = 0:
// one permutation), then `extra` more permutations; h_rates receives the 16 rate elements after the absorption and after each
// extra permutation.  h_state: 17 elements in/out.  All elements 4 words, standard form, < r.
int32_t zp_poseidon_bn254_sponge(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates) {
    return zp_poseidon_bn254_sponge_caps(ctx, h_state, h_blocks, nblocks, extra, h_rates, nullptr);
}

// the same, also handing out the capacity element after every permutation: h_caps u64[max(nblocks, 1) + extra][4] (may be NULL)
int32_t zp_poseidon_bn254_sponge_caps(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates,
                                      uint64_t *h_caps) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_bn254_sponge");
    const P254Table *tb;
    ZP_TRY(table_for(ctx, 17, &tb));
    ZP_ARG(ctx, h_state && h_rates && (h_blocks || nblocks == 0), "null pointer");
    ZP_ARG(ctx, nblocks <= 65536 && extra <= 65536, "too many blocks");
    for (size_t i = 0; i < 17; i++) ZP_ARG(ctx, fr_is_canonical_u64((const u64 *)h_state + 4 * i), "state element not reduced mod r");
    for (size_t i = 0; i < nblocks * 16; i++) ZP_ARG(ctx, fr_is_canonical_u64((const u64 *)h_blocks + 4 * i), "block element not reduced mod r");
    // one upload (state | blocks), one launch walking the blocks on the device, one download (state, rates)
    const size_t nrate = 1 + extra, nperm = (nblocks ? nblocks : 1) + extra, words = (17 + nblocks * 16 + nrate * 16 + nperm) * 4;
    std::vector<u64> h(words, 0);
    memcpy(h.data(), h_state, 17 * 32);
    if (nblocks) memcpy(h.data() + 17 * 4, h_blocks, nblocks * 16 * 32);
    u64 *d = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, words, &d));
    ZP_TRY(zpi_h2d_small(ctx, d, h.data(), (17 + nblocks * 16) * 32));
    u64 *d_caps = d + (17 + nblocks * 16 + nrate * 16) * 4;
    hipLaunchKernelGGL(poseidon254_sponge_kernel, dim3(1), dim3(64), 0, ctx->stream, d, (int)nblocks, (int)extra, h_caps ? d_caps : (u64 *)nullptr, dev_of(ctx, tb));
    ZP_HIP(ctx, hipGetLastError());
    if (h_caps) ZP_TRY(zpi_d2h_small(ctx, h_caps, d_caps, nperm * 32));
    ZP_TRY(zpi_d2h_small(ctx, h_rates, d + (17 + nblocks * 16) * 4, nrate * 16 * 32));
    return zpi_d2h_small(ctx, h_state, d, 17 * 32);
}

// d_tree: u64[nodes][4], leaves first (M of them), then ceil(M/16), ... down to the single root (the last 4 words)
size_t zp_merkle16_nodes(size_t M) {
    size_t n = M, total = M;
    while (n > 1) { n = (n + 15) / 16; total += n; }
    return total;
}

int32_t zp_merkle16_commit_bn254(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, uint64_t *d_tree) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "merkle16_commit_bn254");
    const P254Table *tb;
    ZP_TRY(table_for(ctx, 17, &tb));
    ZP_ARG(ctx, d_cols && d_tree && M >= 1 && W >= 1, "bad arguments");
    const size_t bulk = (size_t)bulk_threshold(ctx);
    if (M >= bulk)
        hipLaunchKernelGGL(p254_bulk_kernel<1>, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, ctx->stream, (const u64 *)d_cols, M, (int)W,
                           (u64 *)d_tree, M, dev_of(ctx, tb));
    else
        hipLaunchKernelGGL(merkle16_leaves_kernel, dim3((unsigned)((M + 2) / 3)), dim3(64), 0, ctx->stream, (const u64 *)d_cols, M, (int)W,
                           (u64 *)d_tree, dev_of(ctx, tb));
    ZP_HIP(ctx, hipGetLastError());
    return zpi_merkle16_levels_bn254(ctx, (u64 *)d_tree, M);
}

// opening of leaf idx: for every level the 16 digests of the group the path passes through (the verifier re-hashes the
// group with its own digest in place): h_path u64[levels][16][4], bottom-up
int32_t zp_merkle16_open_bn254(zp_ctx *ctx, const uint64_t *d_tree, size_t M, size_t idx, uint64_t *h_path) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, d_tree && h_path && idx < M, "bad arguments");
    size_t n = M, off = 0, pos = idx;
    int lvl = 0;
    while (n > 1) {
        const size_t g0 = (pos / 16) * 16;
        uint64_t grp[64];
        memset(grp, 0, sizeof(grp));
        const size_t have = g0 + 16 <= n ? 16 : n - g0;
        ZP_TRY(zpi_d2h_small(ctx, grp, d_tree + (off + g0) * 4, have * 32));
        memcpy(h_path + (size_t)lvl * 64, grp, sizeof(grp));
        off += n;
        n = (n + 15) / 16;
        pos /= 16;
        lvl++;
    }
    return ZP_OK;
}

// the same for nq leaves at once (one gather kernel, one copy): h_paths u64[nq][levels][16][4]
int32_t zp_merkle16_open_batch_bn254(zp_ctx *ctx, const uint64_t *d_tree, size_t M, const uint64_t *h_idx, int32_t nq, uint64_t *h_paths) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, M >= 1 && nq >= 0, "bad sizes");
    int levels = 0;
    for (size_t n = M; n > 1; n = (n + 15) / 16) levels++;
    if (nq == 0 || levels == 0) return ZP_OK;
    ZP_ARG(ctx, d_tree && h_idx && h_paths, "null pointer");
    for (int i = 0; i < nq; i++) ZP_ARG(ctx, h_idx[i] < M, "leaf index out of range");
    const u64 total = (u64)nq * levels * 64;
    u64 *d = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, (size_t)nq + total, &d));
    ZP_TRY(zpi_h2d_small(ctx, d, h_idx, (size_t)nq * 8));
    hipLaunchKernelGGL(merkle16_paths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, (const u64 *)d_tree, (u64)M, levels, d,
                       (int)nq, d + nq);
    ZP_HIP(ctx, hipGetLastError());
    if (total * 8 <= ZP_SMALL_COPY) return zpi_d2h_small(ctx, h_paths, d + nq, total * 8);
    ZP_HIP(ctx, hipMemcpyAsync(h_paths, d + nq, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

}  // extern "C"

// the levels above n digests that already lie at the head of d_tree (u64[zp_merkle16_nodes(n)][4]): what zp_merkle16_commit_bn254 runs after its
// leaves, and what a row-sharded commitment runs on the all-gathered sub-tree digests (csrc/prove.hip: shard_commit_bn)
int32_t zpi_merkle16_levels_bn254(zp_ctx *ctx, u64 *d_tree, size_t n) {
    const P254Table *tb;
    ZP_TRY(table_for(ctx, 17, &tb));
    const size_t bulk = (size_t)bulk_threshold(ctx);
    size_t off = 0;
    while (n > 1) {
        const size_t nn = (n + 15) / 16;
        if (nn >= bulk)
            hipLaunchKernelGGL(p254_bulk_kernel<2>, dim3((unsigned)((nn + 63) / 64)), dim3(64), 0, ctx->stream, (const u64 *)d_tree + off * 4, n, 0,
                               (u64 *)d_tree + (off + n) * 4, nn, dev_of(ctx, tb));
        else
            hipLaunchKernelGGL(merkle16_level_kernel, dim3((unsigned)((nn + 2) / 3)), dim3(64), 0, ctx->stream, (const u64 *)d_tree + off * 4, n,
                               (u64 *)d_tree + (off + n) * 4, nn, dev_of(ctx, tb));
        ZP_HIP(ctx, hipGetLastError());
        off += n;
        n = nn;
    }
    return ZP_OK;
}

// csrc/r1cs.hip: the instances [i0, i0 + count) of the width-17 gadget (d_inst: the instance table of a circuit blob in HBM); the t = 17 tables
// must be installed.  Template numbers checked by the caller: 17 inputs, 8 * 17 + rp S-boxes -> 3 (8 * 17 + rp) + 1 internal wires and rows.
int32_t zpi_r1cs_poseidon17(zp_ctx *ctx, const u64 *d_inst, size_t i0, size_t count, u64 *d_w, unsigned char *d_set, u64 *d_a, u64 *d_b, u64 *d_c,
                            unsigned long long *d_flags, int *rp_out) {
    const P254Table *tb = table_of(ctx, 1);
    ZP_ARG(ctx, tb && tb->t == 17 && tb->d_rc && tb->d_mds, "the width-17 Poseidon-BN254 tables are not installed (zp_set_poseidon_bn254)");
    if (rp_out) *rp_out = tb->rp;
    if (count == 0) return ZP_OK;
    hipLaunchKernelGGL(r1cs_poseidon17_kernel, dim3((unsigned)((count + 2) / 3)), dim3(64), 0, ctx->stream, d_inst, i0, count, d_w, d_set, d_a, d_b, d_c, d_flags,
                       dev_of(ctx, tb));
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
