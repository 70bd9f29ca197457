// Goldilocks NTT / iNTT / LDE for gfx950.
//
// No reference counterpart exists in /root/reference (the prover arithmetic is external to
// eigen-zeth); reference call site served: src/prover/provider.rs:358-377
// (GenChunkProof).  Algorithm: Stockham auto-sort, decimation in frequency, m = ceil(logN/9)
// out-of-place passes of radix R = 2^L (L = 6..9).  Pass i views the column as [R][N/R], a
// workgroup owns a tile of T = 32 consecutive u in [0, N/R):
//     a_r = in[r*(N/R) + u]                               (R segments of T*8 = 256 contiguous bytes)
//     b_k = (sum_r a_r w_R^(rk)) * w_N^(Pprev*k*s)          u = s*Pprev + c,  Pprev = R_1*...*R_(i-1)
//     out[s*Pprev*R + k*Pprev + c] = b_k                    (segments of >= 256 contiguous bytes)
// so natural order goes in and natural order comes out with every global access coalesced in
// 256-byte runs and no separate transpose / bit-reversal pass.  Inside a tile the R-point DFT is
// 2-3 rounds of register radix-2^A (A <= 4, 16 elements per thread, shift-only butterflies) exchanged
// through LDS with an XOR swizzle so that both the row accesses and the transposing copy-out of pass 1 are
// bank-conflict free (MI355X_MICROARCH.md, LDS: ds_read_b64 = 2x32 lanes over 64 banks).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "ctx.hpp"
#include "wave_xchg.hpp"
#include "gl_asm.hpp"
#include "gl_limb.hpp"

// the data of a pass is touched exactly once: non-temporal loads/stores (measured +1 % on the 2^24 bench, same results)
#define ZP_LDG(p) __builtin_nontemporal_load(p)
#define ZP_STG(p, v) __builtin_nontemporal_store(v, p)

namespace {

typedef __attribute__((ext_vector_type(4))) int zp_i32x4;

struct PassArgs {
    const u64 *in;
    u64 *out;
    u64 in_cs, out_cs;  // column strides in elements
    u64 in_valid;       // elements per input column that are real; the rest reads as zero
    const u64 *twl, *twh;
    const u64 *tws;
    const u64 *tw1;        // transposing pass, table form: w^(u k) at [u * R + k] (MODE 3)
    const u64 *csl, *csh;  // coset post-scale tables (last pass of the inverse transform in LDE)
    u64 scale;
    int logn, logPprev, lb, cslb;
    int j0inv;  // w_16(user) = (2^12)^j0, j0inv = j0^-1 mod 16; 0 = root not a power of two path (generic twiddles)
    int flags;  // 1: inter-pass twiddle  2: multiply by `scale`  4: coset post-scale
    int ncols;  // MODE 3: the grid is one-dimensional, columns of one launch
};

__device__ __forceinline__ constexpr int brev(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}
__device__ __forceinline__ constexpr int slot_of(int o, int j, int pos, int A) {
    return ((o >> pos) << (pos + A)) | (j << pos) | (o & ((1 << pos) - 1));
}

template <int A1, int A2, int A3, int LOGT>
struct Geo {
    static constexpr int L = A1 + A2 + A3;
    static constexpr int R = 1 << L;
    static constexpr int T = 1 << LOGT;
    static constexpr int LT = LOGT;
    static constexpr int NT = (R * T) / 16;
    static constexpr int J = 1 + (A2 > 0 ? 1 : 0) + (A3 > 0 ? 1 : 0);
    static constexpr int POS1 = L - A1;
    static constexpr int POS2 = L - A1 - A2;
    static constexpr int AJ = (J == 1) ? A1 : (J == 2 ? A2 : A3);
    // output index k held by slot sigma once all rounds are done
    __device__ static __forceinline__ int kof(int sigma) {
        int k = (sigma >> POS1) & ((1 << A1) - 1);
        if (A2 > 0) k |= ((sigma >> POS2) & ((1 << A2) - 1)) << A1;
        if (A3 > 0) k |= (sigma & ((1 << A3) - 1)) << (A1 + A2);
        return k;
    }
    __device__ static __forceinline__ int sigma_of_k(int k) {
        int s = (k & ((1 << A1) - 1)) << POS1;
        if (A2 > 0) s |= ((k >> A1) & ((1 << A2) - 1)) << POS2;
        if (A3 > 0) s |= (k >> (A1 + A2)) & ((1 << A3) - 1);
        return s;
    }
    // LDS position of (slot sigma, column t).  t is XORed with the low bits of the output index (conflict-free row accesses
    // and transposing copy-out at T >= 16).  With T < 16 one slot is only T*8 < 128 bytes, so the 32 lanes of a half-wave
    // span 32/T slots and those must fall into different 32/T-ths of a 256-byte bank row: the low bits of sigma (inside
    // the last-round field) are XORed with the low bits of the middle field (last-round reads and writes: consecutive lanes
    // = consecutive middle-field values) and, one bit up, with the top-field bits the t-swizzle does not use (transposing
    // copy-out: consecutive lanes = consecutive top-field values, then the middle field steps by one).
    __device__ static __forceinline__ int lpos(int sigma, int t) {
        int s2 = sigma;
        if constexpr (LOGT < 4 && A3 > 0) {
            constexpr int GB = 5 - LOGT;
            s2 ^= ((sigma >> POS2) & ((1 << GB) - 1)) ^ (((sigma >> (POS1 + LOGT)) & ((1 << (A1 - LOGT)) - 1)) << 1);
        }
        return (s2 << LOGT) + (t ^ (kof(sigma) & (T - 1)));
    }
};

__device__ __forceinline__ u64 tw_lookup(const u64 *lo, const u64 *hi, int lb, u64 e) {
    u64 a = lo[e & ((1ULL << lb) - 1)];
    u64 b = hi[e >> lb];
    return gl_mul(a, b);
}


// radix-2^A DIF with the canonical root c = 2^12 (order 16): every twiddle is a shift.
// v[p] receives DFT_c[brev(p)];  the caller maps that to the user's root, w_16 = c^j0:
// DFT_user[k'] = DFT_c[j0*k' mod 2^A]  =>  register p holds user index  k' = j0inv*brev(p) mod 2^A.
__device__ __forceinline__ u64 mul_c16(int e, u64 d) {   // e is a constant after unrolling: the switch folds away
    switch (e) {                                          // x * 2^(12 e): hand-written forms of gl_asm.hpp
        case 0: return d;
        case 1: return gl_shl12<1>(d);
        case 2: return gl_shl12<2>(d);
        case 3: return gl_shl12<3>(d);
        case 4: return gl_shl12<4>(d);
        case 5: return gl_shl12<5>(d);
        case 6: return gl_shl12<6>(d);
        default: return gl_shl12<7>(d);
    }
}
// butterflies two at a time through the hand-scheduled carry chains of gl_asm.hpp (x+y, x-y canonical)
template <int A>
__device__ __forceinline__ void dif_shift(u64 *v) {
    static_assert(A >= 2 && A <= 4, "radix 4, 8 or 16");
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int half = 1 << (A - 1 - s);
#pragma unroll
        for (int t = 0; t < (1 << (A - 1)); t += 2) {
            const int i0 = (t / half) * 2 * half + (t % half), i1 = ((t + 1) / half) * 2 * half + ((t + 1) % half);
            gl_bfly2(v[i0], v[i0 + half], v[i1], v[i1 + half]);
            v[i0 + half] = mul_c16((t % half) * (8 / half), v[i0 + half]);
            v[i1 + half] = mul_c16(((t + 1) % half) * (8 / half), v[i1 + half]);
        }
    }
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain vmcnt, so
// the next tile's global loads (and the previous tile's stores) stay in flight across it.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// twr: LDS copy of w_(2^(POS+A))^e (TWSH = 0) or, for the radix-2^11 / 2^12 passes whose LDS is all tile, the global
// table of w_4096^e (TWSH = 12 - (POS + A)); 32 KiB, read by every workgroup, L2 / L1 resident
template <typename G, int A, int POS, int ANEXT, int POSNEXT, int TWSH>
__device__ __forceinline__ void exchange2(u64 *v, u64 *lds, const u64 *twr, int j0inv, int tid) {
    constexpr int GR = 16 >> A;
#pragma unroll
    for (int g = 0; g < GR; g++) {
        const int gamma = g * G::NT + tid;
        const int t = gamma & (G::T - 1), o = gamma >> G::LT;
        const int rho = o & ((1 << POS) - 1);
        // w_(2^(POS+A))^(kj*rho) from the LDS copy; p = 0 needs none, the rest go two products at a time
        {
            const int kj1 = (j0inv * brev(1, A)) & ((1 << A) - 1);
            v[g * (1 << A) + 1] = gl_mul1(v[g * (1 << A) + 1], twr[(kj1 * rho) << TWSH]);
        }
#pragma unroll
        for (int p = 2; p < (1 << A); p += 2) {
            const int kja = (j0inv * brev(p, A)) & ((1 << A) - 1), kjb = (j0inv * brev(p + 1, A)) & ((1 << A) - 1);
            gl_mul2(v[g * (1 << A) + p], twr[(kja * rho) << TWSH], v[g * (1 << A) + p + 1], twr[(kjb * rho) << TWSH]);
        }
#pragma unroll
        for (int p = 0; p < (1 << A); p++) {
            const int kj = (j0inv * brev(p, A)) & ((1 << A) - 1);  // wave-uniform
            lds[G::lpos(slot_of(o, kj, POS, A), t)] = v[g * (1 << A) + p];
        }
    }
    lds_barrier();
    constexpr int GN = 16 >> ANEXT;
#pragma unroll
    for (int g = 0; g < GN; g++) {
        const int gamma = g * G::NT + tid;
        const int t = gamma & (G::T - 1), o = gamma >> G::LT;
#pragma unroll
        for (int j = 0; j < (1 << ANEXT); j++)
            v[g * (1 << ANEXT) + j] = lds[G::lpos(slot_of(o, j, POSNEXT, ANEXT), t)];
    }
}

// ---- the same two building blocks on four signed 24-bit-position limbs (gl_limb.hpp): butterflies are carry-free 32-bit adds, every
// twiddle 2^(24 j) a renaming of limbs, and the canonical 64-bit value comes back inside the twiddle product that follows a round
template <int E>
__device__ __forceinline__ gl_l4 mul_c16_l4(const gl_l4 &d) { return gl_l4_mul_c16<E>(d); }
__device__ __forceinline__ gl_l4 mul_c16_l4(int e, const gl_l4 &d) {   // e is a constant after unrolling
    switch (e) {
        case 0: return d;
        case 1: return gl_l4_mul_c16<1>(d);
        case 2: return gl_l4_mul_c16<2>(d);
        case 3: return gl_l4_mul_c16<3>(d);
        case 4: return gl_l4_mul_c16<4>(d);
        case 5: return gl_l4_mul_c16<5>(d);
        case 6: return gl_l4_mul_c16<6>(d);
        default: return gl_l4_mul_c16<7>(d);
    }
}
// the A butterfly levels on limb-form values: x[p] <- DFT_c[brev(p)], limbs grow by one bit per level
template <int A>
__device__ __forceinline__ void dif_levels_l4(gl_l4 *x) {
    static_assert(A >= 2 && A <= 4, "radix 4, 8 or 16");
#pragma unroll
    for (int s = 0; s < A; s++) {
        const int half = 1 << (A - 1 - s);
#pragma unroll
        for (int t = 0; t < (1 << (A - 1)); t++) {
            const int i0 = (t / half) * 2 * half + (t % half);
            const gl_l4 a = x[i0], b = x[i0 + half];
            x[i0] = gl_l4_add(a, b);
            x[i0 + half] = mul_c16_l4((t % half) * (8 / half), gl_l4_sub(a, b));
        }
    }
}
// x[p] receives DFT_c[brev(p)] of the canonical values v[0 .. 2^A), |limbs| < 2^(24 + A)
template <int A>
__device__ __forceinline__ void dif_shift_l4(const u64 *v, gl_l4 *x) {
#pragma unroll
    for (int i = 0; i < (1 << A); i++) x[i] = gl_l4_from(v[i]);
    dif_levels_l4<A>(x);
}
// the LDS record of a factor: balanced words of w B^i, i < 4 (32 bytes)
__device__ __forceinline__ gl_w4 w4_load(const gl_w4 *p) {
    const zp_i32x4 a = *(const zp_i32x4 *)p->lo, b = *(const zp_i32x4 *)p->hi;
    gl_w4 w;
    w.lo[0] = a.x; w.lo[1] = a.y; w.lo[2] = a.z; w.lo[3] = a.w;
    w.hi[0] = b.x; w.hi[1] = b.y; w.hi[2] = b.z; w.hi[3] = b.w;
    return w;
}
__device__ __forceinline__ void w4_store(gl_w4 *p, const u64 w0, const u64 w1, const u64 w2, const u64 w3) {
    i32 l[4], h[4];
    gl_l4_balance(w0, l[0], h[0]);
    gl_l4_balance(w1, l[1], h[1]);
    gl_l4_balance(w2, l[2], h[2]);
    gl_l4_balance(w3, l[3], h[3]);
    zp_i32x4 a = {l[0], l[1], l[2], l[3]}, b = {h[0], h[1], h[2], h[3]};
    *(zp_i32x4 *)p->lo = a;
    *(zp_i32x4 *)p->hi = b;
}
// twr4: LDS records of w_(2^(POS+A))^e
template <typename G, int A, int POS, int ANEXT, int POSNEXT>
__device__ __forceinline__ void exchange2_l4(const gl_l4 *x, u64 *v, u64 *lds, const gl_w4 *twr4, int j0inv, int tid) {
    constexpr int GR = 16 >> A;
#pragma unroll
    for (int g = 0; g < GR; g++) {
        const int gamma = g * G::NT + tid;
        const int t = gamma & (G::T - 1), o = gamma >> G::LT;
        const int rho = o & ((1 << POS) - 1);
#pragma unroll
        for (int p = 0; p < (1 << A); p++) {
            const int kj = (j0inv * brev(p, A)) & ((1 << A) - 1);  // wave-uniform
            const u64 y = p == 0 ? gl_l4_canon(x[g * (1 << A)]) : gl_l4_mul(x[g * (1 << A) + p], w4_load(twr4 + kj * rho));
            lds[G::lpos(slot_of(o, kj, POS, A), t)] = y;
        }
    }
    lds_barrier();
    constexpr int GN = 16 >> ANEXT;
#pragma unroll
    for (int g = 0; g < GN; g++) {
        const int gamma = g * G::NT + tid;
        const int t = gamma & (G::T - 1), o = gamma >> G::LT;
#pragma unroll
        for (int j = 0; j < (1 << ANEXT); j++)
            v[g * (1 << ANEXT) + j] = lds[G::lpos(slot_of(o, j, POSNEXT, ANEXT), t)];
    }
}

// Pass kernel, second generation.
//  * A workgroup walks `tiles_per_wg` consecutive tiles of one column.  ALL vector global loads of
//    tile i+1 (16 data elements per lane + the few twiddle-table entries it needs) are issued at the
//    top of iteration i and consumed in iteration i+1; the body of an iteration touches only
//    registers and LDS, and barriers order LDS only.  vmcnt is an in-order counter, so this is what
//    keeps HBM loads (and the previous tile's stores) in flight under the integer work.
//  * butterflies use the canonical 16-th root 2^12 (shifts only); the user's root enters through
//    j0inv as a renaming of outputs.
//  * TRANSPOSE=false (passes 2..m, Pprev >= T): the tile shares s; the inter-pass twiddle w^(e0*k)
//    (times 1/N and the k-part of a coset power on a last pass) is a per-tile table of R entries in
//    LDS -> one multiplication per element.
//  * TRANSPOSE=true (pass 1, Pprev = 1): s = u varies with the lane; w^(u*k) is a per-lane geometric
//    chain walked in the order the shift butterflies deliver outputs: k_j = jr*i mod 2^A, so
//    h^(k_j) = (h^jr)^i * (h^-(2^A))^floor(jr*i/2^A).
//    Table form (MODE 3): the N factors w^(u*k) of a transform are the same for every column, so they are precomputed once per
//    plan in the order of the transposing copy-out ([u][k]: 8 bytes per element, coalesced) and multiplied in while the tile
//    leaves LDS -- one product per element instead of the chain's two and none of its exponent arithmetic.  What makes the
//    extra 8 bytes per element affordable is that they are read from HBM once per 8 columns: the grid is one-dimensional and
//    ordered so that the columns of one tile run next to each other on ONE XCD (workgroup id % 8 picks the XCD), and the
//    table loads are cacheable, so 7 of 8 reads are hits in that XCD's L2.
// MODE (non-transposing passes): 0 = plain last pass, 1 = multiply by the per-tile table,
// 2 = table and the per-lane part of a coset power (last pass of the inverse transform in an LDE)
// LIMB: the register butterflies and the twiddle products that follow them on the limb form of gl_limb.hpp (16 values x 4 limbs live
// between the rounds: 3 waves per SIMD, 168 VGPRs); the per-tile table and the LDS copies of the inter-round twiddles hold 32-byte
// records (the balanced words of w, w 2^24, w 2^48, w 2^72).  Not for the per-lane twiddle chain of a first pass without its table.
template <int A1, int A2, int A3, int LOGT, bool TRANSPOSE, bool PADDED, int MODE, bool BIG = false, bool LIMB = false>
__global__ void __launch_bounds__((1 << (A1 + A2 + A3 + LOGT)) / 16, LIMB ? 3 : 4)   // 4 waves per SIMD: 128 VGPRs, of which v116..v127 are the scratch window of gl_asm.hpp
ntt_pass2_kernel(PassArgs a, int tiles_per_wg) {
    using G = Geo<A1, A2, A3, LOGT>;
    constexpr int L = G::L, R = G::R, T = G::T, NT = G::NT, AJ = G::AJ;
    constexpr int GRJ = 16 >> AJ;
    constexpr int R2 = 1 << (A2 + A3);
    constexpr bool HAS_TAB = !TRANSPOSE && MODE >= 1;
    constexpr bool TWR_LDS = L <= 10;          // radix 2^11 / 2^12: the 128 KiB tile leaves no room, twiddles come from L2
    constexpr int TPL = (R + NT - 1) / NT;     // per-tile table entries a lane prepares
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    constexpr int REC = LIMB ? 4 : 1;          // u64 per table entry (LIMB: a 32-byte gl_w4 record)
    u64 *tab = lds + R * T;                    // R entries: per-tile inter-pass twiddles (HAS_TAB)
    u64 *twr1 = tab + (HAS_TAB ? R * REC : 0); // R entries:  w_R^e      (inter-round twiddles, first exchange)
    u64 *twr2 = twr1 + R * REC;                // R2 entries: w_(R2)^e   (second exchange, 3-round passes)
    constexpr bool TW1 = TRANSPOSE && MODE == 3;
    static_assert(!LIMB || (TWR_LDS && (TW1 || !TRANSPOSE)), "limb form: LDS twiddle copies, no per-lane twiddle chain");
    // MODE 3: one-dimensional grid; workgroup id -> (XCD x = id % 8, j = id / 8): column = j % ncols, tile group = (j / ncols) * 8 + x
    u64 col = blockIdx.y, wg_lin = blockIdx.x;
    if constexpr (TW1) {
        const u64 x = blockIdx.x & 7u, j = blockIdx.x >> 3;
        col = j % (u64)a.ncols;
        wg_lin = (j / (u64)a.ncols) * 8 + x;
    }
    const int logNR = a.logn - L;
    const int logP = a.logPprev;
    const u64 N = 1ULL << a.logn;
    const u64 *src = a.in + col * a.in_cs;
    u64 *dst = a.out + col * a.out_cs;
    u64 v[16], vn[16];
    // twiddle-table entries fetched one tile ahead (raw lo/hi halves)
    constexpr int NTW = TW1 ? 1 : TRANSPOSE ? (GRJ + 2) : (MODE == 2 ? TPL + 1 : TPL);
    u64 twl_c[NTW], twh_c[NTW], twl_n[NTW], twh_n[NTW];

    // Addressing: row j of a lane's 16 loads is  (uniform: column base + (j << POS1) rows + u0)  +  (lane: slot0 rows + t).
    // With the lane part as a 32-bit BYTE offset the loads take the scalar-base form (global_load v, voff, s[base]) and the
    // 64-bit address arithmetic -- three VALU instructions per load when written naively -- is scalar work.  The byte
    // offset fits 32 bits up to 2^28 rows; larger transforms are the BIG instantiation with 64-bit lane offsets.
    constexpr bool small_n = !BIG;
    auto fetch_tile = [&](u64 *r, u64 *tl, u64 *th, u64 u0, int tid) {
        constexpr int GR = 16 >> A1;
        const u64 valid_rows = a.in_valid >> logNR;     // PADDED: rows >= valid_rows read as zero (in_valid is a multiple of N/R)
#pragma unroll
        for (int g = 0; g < GR; g++) {
            const int gamma = g * NT + tid;
            const int t = gamma & (T - 1), o = gamma >> LOGT;
            const u32 slot0 = (u32)slot_of(o, 0, G::POS1, A1);
            const u32 lane_bytes = (((u32)slot0 << logNR) + (u32)t) << 3;
            const u64 lane_el = ((u64)slot0 << logNR) + (u64)t;
#pragma unroll
            for (int j = 0; j < (1 << A1); j++) {
                const u64 *rowp = src + (((u64)j << G::POS1) << logNR) + u0;      // wave-uniform
                const u64 *ptr;
                if constexpr (small_n) ptr = (const u64 *)((const char *)rowp + lane_bytes);
                else ptr = rowp + lane_el;
                // short runs (T < 16) want the L2 to merge neighbouring tiles' pieces of a line: plain, cacheable loads
                u64 val;
                if constexpr (PADDED) val = (u64)(slot0 + ((u32)j << G::POS1)) < valid_rows ? (LOGT < 4 ? *ptr : ZP_LDG(ptr)) : 0ULL;
                else val = LOGT < 4 ? *ptr : ZP_LDG(ptr);
                r[g * (1 << A1) + j] = val;
            }
        }
        const u64 lm = (1ULL << a.lb) - 1;
        if constexpr (TW1) {
            (void)tl; (void)th; (void)lm;
        } else if constexpr (TRANSPOSE) {
            const int t = tid & (T - 1);
            const u64 u = u0 + t;
            const int jr = a.j0inv & ((1 << AJ) - 1);
            u64 e[NTW];
#pragma unroll
            for (int g = 0; g < GRJ; g++) e[g] = u * (u64)G::kof(((g * NT + tid) >> LOGT) << AJ);   // w^(u*klow_g)
            e[GRJ] = ((u << (L - AJ)) * (u64)jr) & (N - 1);                                       // h^jr
            e[GRJ + 1] = (N - ((u << L) & (N - 1))) & (N - 1);                                    // h^-(2^AJ) = w^-(u*R)
#pragma unroll
            for (int i = 0; i < NTW; i++) { tl[i] = a.twl[e[i] & lm]; th[i] = a.twh[e[i] >> a.lb]; }
        } else {
            if constexpr (MODE >= 1) {
                const u64 e0 = (u0 >> logP) << logP;
#pragma unroll
                for (int j = 0; j < TPL; j++) {
                    const u64 e = (a.flags & 1) ? e0 * (u64)((tid + j * NT) & (R - 1)) : 0;  // entry k = tid + j*NT (k < R used)
                    tl[j] = a.twl[e & lm];
                    th[j] = a.twh[e >> a.lb];
                }
            }
            if constexpr (MODE == 2) {
                const u64 c = (u0 + (tid & (T - 1))) & ((1ULL << logP) - 1);
                tl[TPL] = a.csl[c & ((1ULL << a.cslb) - 1)];
                th[TPL] = a.csh[c >> a.cslb];
            }
        }
    };

    // Runs shorter than a 128-byte line (T < 16) are shared between neighbouring tiles: hand neighbours to workgroups
    // of the same XCD (blockIdx % 8 -- a speed hint only, MI355X_MICROARCH.md "Workgroup dispatch") so that the other
    // parts of a line are L2 hits; measured on the bare access pattern: 2.7 -> 4.8 TB/s (profiles/r2_ubench_mem.txt)
    u64 wg = wg_lin;
    if constexpr (LOGT < 4 && !TW1) {
        if ((gridDim.x & 7u) == 0) wg = (u64)(blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    }
    const u64 tile0 = wg * tiles_per_wg;
    fetch_tile(v, twl_c, twh_c, tile0 << LOGT, threadIdx.x);
    // once per workgroup: inter-round twiddles into LDS, tile-independent factors into registers
    u64 kfac[TPL];
    {
        const int tid0 = threadIdx.x;
        if constexpr (LIMB) {
            // w 2^(24 i) = w_R^(e + i rot): 2^12 = w_16^(j0inv), so 2^24 = w_R^(j0inv R / 8) -- four entries of the same table
            for (int e = tid0; e < R; e += NT) {
                const int rot = (a.j0inv * (R / 8)) & (R - 1);
                w4_store((gl_w4 *)twr1 + e, a.tws[e << (12 - L)], a.tws[((e + rot) & (R - 1)) << (12 - L)],
                         a.tws[((e + 2 * rot) & (R - 1)) << (12 - L)], a.tws[((e + 3 * rot) & (R - 1)) << (12 - L)]);
            }
            if constexpr (G::J >= 3) {
                constexpr int L2 = A2 + A3;
                for (int e = tid0; e < R2; e += NT) {
                    const int rot = (a.j0inv * (R2 / 8)) & (R2 - 1);
                    w4_store((gl_w4 *)twr2 + e, a.tws[e << (12 - L2)], a.tws[((e + rot) & (R2 - 1)) << (12 - L2)],
                             a.tws[((e + 2 * rot) & (R2 - 1)) << (12 - L2)], a.tws[((e + 3 * rot) & (R2 - 1)) << (12 - L2)]);
                }
            }
        } else if constexpr (TWR_LDS) {
            if (tid0 < R) twr1[tid0] = a.tws[tid0 << (12 - L)];
            if constexpr (G::J >= 3) {
                if (tid0 < R2) twr2[tid0] = a.tws[tid0 << (12 - (A2 + A3))];
            }
        }
#pragma unroll
        for (int j = 0; j < TPL; j++) {
            kfac[j] = 1;
            if constexpr (HAS_TAB) {
                if (a.flags & 2) kfac[j] = a.scale;
                if constexpr (MODE == 2) {
                    const u64 ek = (u64)((tid0 + j * NT) & (R - 1)) << logP;
                    kfac[j] = gl_mul(kfac[j], gl_mul(a.csl[ek & ((1ULL << a.cslb) - 1)], a.csh[ek >> a.cslb]));
                }
            }
        }
    }
    lds_barrier();  // twr1/twr2 are read by other lanes before the first exchange barrier
    // drain the prologue loads here so that the loop is entered with nothing pending: the waitcnt
    // insertion then needs no vmcnt wait inside the body (one there would also drain the prefetch)
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
#pragma clang loop unroll(disable)
    for (int it = 0; it < tiles_per_wg; it++) {
        const u64 u0 = (tile0 + it) << LOGT;
        const bool more = it + 1 < tiles_per_wg;
        // opaque copy of the lane id: keeps the compiler from hoisting every per-lane address of the
        // loop body out of the loop (that costs > 100 VGPRs and spills)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        if (more) fetch_tile(vn, twl_n, twh_n, u0 + T, tid);

        const u64 s = TRANSPOSE ? 0 : (u0 >> logP);
        if constexpr (HAS_TAB) {
#pragma unroll
            for (int j = 0; j < TPL; j++)
                if (tid + j * NT < R) {
                    const u64 w = gl_mul(gl_mul(twl_c[j], twh_c[j]), kfac[j]);
                    if constexpr (LIMB) w4_store((gl_w4 *)tab + tid + j * NT, w, gl_shl12<2>(w), gl_shl12<4>(w), gl_shl12<6>(w));
                    else tab[tid + j * NT] = w;
                }
        }
        gl_l4 x[LIMB ? 16 : 1];                // LIMB: the outputs of a round (the last one: x, else v)
        if constexpr (LIMB) {
#pragma unroll
            for (int g = 0; g < (16 >> A1); g++) dif_shift_l4<A1>(v + g * (1 << A1), x + g * (1 << A1));
            exchange2_l4<G, A1, G::POS1, A2, G::POS2>(x, v, lds, (const gl_w4 *)twr1, a.j0inv, tid);
#pragma unroll
            for (int g = 0; g < (16 >> A2); g++) dif_shift_l4<A2>(v + g * (1 << A2), x + g * (1 << A2));
            if constexpr (G::J >= 3) {
                exchange2_l4<G, A2, G::POS2, A3, 0>(x, v, lds, (const gl_w4 *)twr2, a.j0inv, tid);
#pragma unroll
                for (int g = 0; g < (16 >> A3); g++) dif_shift_l4<A3>(v + g * (1 << A3), x + g * (1 << A3));
            }
        } else {
#pragma unroll
        for (int g = 0; g < (16 >> A1); g++) dif_shift<A1>(v + g * (1 << A1));
        exchange2<G, A1, G::POS1, A2, G::POS2, TWR_LDS ? 0 : 12 - L>(v, lds, TWR_LDS ? twr1 : a.tws, a.j0inv, tid);
#pragma unroll
        for (int g = 0; g < (16 >> A2); g++) dif_shift<A2>(v + g * (1 << A2));
        if constexpr (G::J >= 3) {
            exchange2<G, A2, G::POS2, A3, 0, TWR_LDS ? 0 : 12 - (A2 + A3)>(v, lds, TWR_LDS ? twr2 : a.tws, a.j0inv, tid);
#pragma unroll
            for (int g = 0; g < (16 >> A3); g++) dif_shift<A3>(v + g * (1 << A3));
        }
        }
#pragma unroll
        for (int g = 0; g < GRJ; g++) {
            const int gamma = g * NT + tid;
            const int t = gamma & (T - 1), o = gamma >> LOGT;
            const int klow = G::kof(o << AJ);
            if constexpr (TW1) {
                const int jr = a.j0inv & ((1 << AJ) - 1);
#pragma unroll
                for (int i = 0; i < (1 << AJ); i++) {
                    const int kj = (jr * i) & ((1 << AJ) - 1);
                    if constexpr (LIMB) lds[G::lpos(slot_of(o, kj, 0, AJ), t)] = gl_l4_canon(x[g * (1 << AJ) + brev(i, AJ)]);
                    else lds[G::lpos(slot_of(o, kj, 0, AJ), t)] = v[g * (1 << AJ) + brev(i, AJ)];
                }
            } else if constexpr (TRANSPOSE) {
                const int jr = a.j0inv & ((1 << AJ) - 1);  // odd, < 2^AJ: floor(jr*i/2^AJ) steps by 0 or 1
                u64 w = gl_mul(twl_c[g], twh_c[g]);
                const u64 hj = gl_mul(twl_c[GRJ], twh_c[GRJ]);
                const u64 hjd = gl_mul(hj, gl_mul(twl_c[GRJ + 1], twh_c[GRJ + 1]));
#pragma unroll
                for (int i = 0; i < (1 << AJ); i++) {
                    const int p = brev(i, AJ);
                    const int kj = (jr * i) & ((1 << AJ) - 1);
                    u64 x = v[g * (1 << AJ) + p];
                    if (i + 1 < (1 << AJ)) {
                        const bool wrap = ((jr * (i + 1)) >> AJ) != ((jr * i) >> AJ);  // wave-uniform
                        u64 wn = w;
                        gl_mul2(x, w, wn, wrap ? hjd : hj);
                        w = wn;
                    } else {
                        x = gl_mul1(x, w);
                    }
                    lds[G::lpos(slot_of(o, kj, 0, AJ), t)] = x;
                }
            } else {
                // store address = (uniform: s-block + the tile's first column + k-step)  +  (lane: klow rows + t), as for the loads
                u64 *const sbase = dst + (s << (logP + L)) + (u0 & ((1ULL << logP) - 1));
                const u32 lane_sb = (((u32)klow << logP) + (u32)t) << 3;
                const u64 lane_se = ((u64)klow << logP) + (u64)t;
                u64 cw = 1;
                if constexpr (MODE == 2) cw = gl_mul(twl_c[TPL], twh_c[TPL]);  // shift^c, c < Pprev
#pragma unroll
                for (int p = 0; p < (1 << AJ); p += 2) {
                    const int kja = (a.j0inv * brev(p, AJ)) & ((1 << AJ) - 1), kjb = (a.j0inv * brev(p + 1, AJ)) & ((1 << AJ) - 1);
                    const int ka = klow + (kja << (L - AJ)), kb = klow + (kjb << (L - AJ));
                    u64 xa, xb;
                    if constexpr (LIMB) {
                        if constexpr (MODE >= 1) {
                            xa = gl_l4_mul(x[g * (1 << AJ) + p], w4_load((const gl_w4 *)tab + ka));
                            xb = gl_l4_mul(x[g * (1 << AJ) + p + 1], w4_load((const gl_w4 *)tab + kb));
                        } else {
                            xa = gl_l4_canon(x[g * (1 << AJ) + p]);
                            xb = gl_l4_canon(x[g * (1 << AJ) + p + 1]);
                        }
                    } else {
                        xa = v[g * (1 << AJ) + p];
                        xb = v[g * (1 << AJ) + p + 1];
                        if constexpr (MODE >= 1) gl_mul2(xa, tab[ka], xb, tab[kb]);
                    }
                    if constexpr (MODE == 2) gl_mul2(xa, cw, xb, cw);
                    u64 *const ua = sbase + ((u64)(kja << (L - AJ)) << logP), *const ub = sbase + ((u64)(kjb << (L - AJ)) << logP);   // uniform
                    u64 *pa, *pb;
                    if constexpr (small_n) { pa = (u64 *)((char *)ua + lane_sb); pb = (u64 *)((char *)ub + lane_sb); }
                    else { pa = ua + lane_se; pb = ub + lane_se; }
                    if constexpr (LOGT < 4) {
                        *pa = xa;
                        *pb = xb;
                    } else {
                        ZP_STG(pa, xa);
                        ZP_STG(pb, xb);
                    }
                }
            }
        }
        if constexpr (TW1) {
            // v is dead (the tile sits in LDS): its registers take the 16 factors of the elements this lane copies out
            const u64 *twb = a.tw1 + (u0 << L);
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = twb[i * NT + tid];       // cacheable: the other columns of this tile hit in L2
            lds_barrier();
            u64 *blk = dst + (u0 << L);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int ia = i * NT + tid, ib = (i + 1) * NT + tid;
                u64 xa = lds[G::lpos(G::sigma_of_k(ia & ((1 << L) - 1)), ia >> L)];
                u64 xb = lds[G::lpos(G::sigma_of_k(ib & ((1 << L) - 1)), ib >> L)];
                gl_mul2(xa, v[i], xb, v[i + 1]);
                ZP_STG(&blk[ia], xa);
                ZP_STG(&blk[ib], xb);
            }
        } else if constexpr (TRANSPOSE) {
            lds_barrier();
            u64 *blk = dst + (u0 << L);
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int idx = i * NT + tid;
                const int t2 = idx >> L, k = idx & ((1 << L) - 1);
                ZP_STG(&blk[idx], lds[G::lpos(G::sigma_of_k(k), t2)]);
            }
        }
        if (more) {
            lds_barrier();  // LDS (tile + table) is reused by the next tile
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = vn[i];
#pragma unroll
            for (int i = 0; i < NTW; i++) { twl_c[i] = twl_n[i]; twh_c[i] = twh_n[i]; }
        }
    }
}

// ---- the seam of an extension (blow-up 2): the LAST pass of the inverse transform and the FIRST pass of the zero-padded forward transform
// in one kernel.  The inverse's last radix-256 pass delivers a tile of coefficients { k N/256 + c : k < 256, c in 16 consecutive } -- and
// the forward transform of 2N points starts with a radix-256 pass over { r 2N/256 + u : r < 128 } (rows 128.. are the zero padding): the
// same index set, k = 2 r + (u >> log(N/256)), c = u mod N/256.  So one inverse tile IS two forward tiles (k even / k odd), and the
// scaled coefficients never have to leave the CU: they are handed over through the tile's LDS (the 128 non-zero rows of both forward
// tiles = the 4 096 elements of the inverse tile), each lane picks up its 2 x 8 non-zero inputs, and the two forward tiles go through
// the first-pass body (first butterfly level of a half-zero register file: a copy and a shift).  Per column this saves the write and
// the read of the coefficient buffer (16 N of 136 N bytes; 8 N when the caller wants the coefficients: `coef`), and one launch.
// Inverse side = ntt_pass2_kernel<4,4,0,4,false,false,2>, forward side = <4,4,0,4,true,true,3>: the same values, bit for bit.
struct SeamArgs {
    const u64 *in;          // input of the inverse transform's last pass, columns of N
    u64 *out;               // output of the forward transform's first pass, columns of 2N
    u64 *coef;              // scaled coefficients c_i shift^i, columns of N (may be null)
    const u64 *tws_i, *tws_f;   // w_4096^e of the inverse / the forward plan
    const u64 *tw1;         // the forward plan's first-pass table, [u * 256 + k]
    const u64 *csl, *csh;   // coset table of the inverse transform's post-scale
    u64 scale;              // 1 / N
    int logn, cslb, j0inv_i, j0inv_f, ncols;
};
// No prefetch of the next tile (unlike ntt_pass2_kernel): a tile is three bodies of integer work per load, the other workgroups of the CU cover
// the one exposed load latency, and the 32 registers it would hold keep the kernel at 128 VGPRs = four waves per SIMD (with them it spilled).
// (The limb form of gl_limb.hpp was tried here too, this kernel being bound by its integer work: 64 more live registers -- 0.77 ms at two waves
// per SIMD, 1.37 ms spilling at three, against 0.70 ms: profiles/r4_lde_seam_ab.txt.  Removed.)
__global__ void __launch_bounds__(256, 4) lde_seam_kernel(SeamArgs a, int tiles_per_wg) {
    using G = Geo<4, 4, 0, 4>;
    constexpr int L = 8, R = 256, T = 16, NT = 256, LOGT = 4;
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    u64 *tab = lds + R * T;        // 1/N shift^(k N/256)
    u64 *twr_i = tab + R;          // inter-round twiddles of the inverse pass
    u64 *twr_f = twr_i + R;        // ... of the forward pass
    const u64 x8 = blockIdx.x & 7u, jj = blockIdx.x >> 3;
    const u64 col = jj % (u64)a.ncols, wg = (jj / (u64)a.ncols) * 8 + x8;
    const int logNR = a.logn - L;                      // log(N / 256): the inverse pass's Pprev, and the split of u
    const u64 N = 1ULL << a.logn;
    const u64 *src = a.in + col * N;
    u64 *dst = a.out + col * (2 * N);
    u64 *cdst = a.coef ? a.coef + col * N : nullptr;
    u64 v[16];
    u64 cw_c[2];                                       // raw halves of shift^c for this lane's column c
    auto fetch_tile = [&](u64 *r, u64 *cwr, u64 u0, int tid) {
        const int t = tid & (T - 1), o = tid >> LOGT;
        const u32 slot0 = (u32)slot_of(o, 0, G::POS1, 4);
        const u32 lane_bytes = (((u32)slot0 << logNR) + (u32)t) << 3;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u64 *rowp = src + (((u64)j << G::POS1) << logNR) + u0;      // wave-uniform
            r[j] = ZP_LDG((const u64 *)((const char *)rowp + lane_bytes));
        }
        const u64 c = u0 + (u64)t;
        cwr[0] = a.csl[c & ((1ULL << a.cslb) - 1)];
        cwr[1] = a.csh[c >> a.cslb];
    };
    const u64 tile0 = wg * tiles_per_wg;
    {
        const int tid0 = threadIdx.x;
        twr_i[tid0] = a.tws_i[tid0 << (12 - L)];
        twr_f[tid0] = a.tws_f[tid0 << (12 - L)];
        const u64 ek = (u64)tid0 << logNR;
        tab[tid0] = gl_mul(a.scale, gl_mul(a.csl[ek & ((1ULL << a.cslb) - 1)], a.csh[ek >> a.cslb]));
    }
    lds_barrier();
#pragma clang loop unroll(disable)
    for (int it = 0; it < tiles_per_wg; it++) {
        const u64 u0 = (tile0 + it) << LOGT;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        fetch_tile(v, cw_c, u0, tid);
        const int t = tid & (T - 1), o = tid >> LOGT;
        // ---- inverse transform, last pass (radix 256, plain last pass of the plan: no inter-pass twiddle; 1/N and the coset power)
        dif_shift<4>(v);
        exchange2<G, 4, G::POS1, 4, G::POS2, 0>(v, lds, twr_i, a.j0inv_i, tid);
        dif_shift<4>(v);
        const u64 cw = gl_mul(cw_c[0], cw_c[1]);       // shift^c
        lds_barrier();                                  // the exchange's reads are done: the tile becomes the hand-over buffer
#pragma unroll
        for (int p = 0; p < 16; p += 2) {
            const int kja = (a.j0inv_i * brev(p, 4)) & 15, kjb = (a.j0inv_i * brev(p + 1, 4)) & 15;
            const int ka = o + (kja << 4), kb = o + (kjb << 4);           // klow = o for <4,4,0>
            u64 xa = v[p], xb = v[p + 1];
            gl_mul2(xa, tab[ka], xb, tab[kb]);
            gl_mul2(xa, cw, xb, cw);
            if (cdst) {
                ZP_STG((u64 *)((char *)(cdst + ((u64)ka << logNR) + u0) + (t << 3)), xa);
                ZP_STG((u64 *)((char *)(cdst + ((u64)kb << logNR) + u0) + (t << 3)), xb);
            }
            // forward tile par = k & 1, row r = k >> 1: [par][r][t]
            lds[(((ka & 1) << 7) + (ka >> 1)) * T + t] = xa;
            lds[(((kb & 1) << 7) + (kb >> 1)) * T + t] = xb;
        }
        lds_barrier();
        // this lane's non-zero inputs of both forward tiles: rows r = 16 j + o, j < 8 (rows 128.. are the padding)
        u64 f1[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            v[j] = lds[((j << 4) + o) * T + t];
            f1[j] = lds[(128 + (j << 4) + o) * T + t];
        }
        lds_barrier();
        // ---- forward transform of 2N points, first pass, tiles u0f = par * N/256 + u0
#pragma unroll 1
        for (int par = 0; par < 2; par++) {
            if (par) {
#pragma unroll
                for (int j = 0; j < 8; j++) v[j] = f1[j];
            }
            // first butterfly level of a half-zero register file: (x + 0, (x - 0) 2^(12 j)); then two radix-8 halves
#pragma unroll
            for (int j = 1; j < 8; j++) v[8 + j] = mul_c16(j, v[j]);
            v[8] = v[0];
            dif_shift<3>(v);
            dif_shift<3>(v + 8);
            exchange2<G, 4, G::POS1, 4, G::POS2, 0>(v, lds, twr_f, a.j0inv_f, tid);
            dif_shift<4>(v);
            {
                const int jr = a.j0inv_f & 15;
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int kj = (jr * i) & 15;
                    lds[G::lpos(slot_of(o, kj, 0, 4), t)] = v[brev(i, 4)];
                }
            }
            const u64 u0f = ((u64)par << logNR) + u0;
            const u64 *twb = a.tw1 + (u0f << L);
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = twb[i * NT + tid];       // cacheable: the other columns of this tile hit in L2
            lds_barrier();
            u64 *blk = dst + (u0f << L);
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const int ia = i * NT + tid, ib = (i + 1) * NT + tid;
                u64 xa = lds[G::lpos(G::sigma_of_k(ia & ((1 << L) - 1)), ia >> L)];
                u64 xb = lds[G::lpos(G::sigma_of_k(ib & ((1 << L) - 1)), ib >> L)];
                gl_mul2(xa, v[i], xb, v[i + 1]);
                ZP_STG(&blk[ia], xa);
                ZP_STG(&blk[ib], xb);
            }
            lds_barrier();      // the tile is reused by the second forward tile / the next inverse tile
        }
    }
}

// ---- small transforms (N <= 4096): one workgroup per column, radix-2 DIF stages in LDS
struct SmallArgs {
    const u64 *in;
    u64 *out;
    u64 in_cs, out_cs, in_valid;
    const u64 *tws;
    const u64 *csl, *csh;
    u64 scale;
    int logn, cslb, flags;
};

// one DIF stage of half 2^LH inside a wave: lane l holds element l of a 64-element block; the lane with bit LH clear keeps the sum, its
// partner the twiddled difference.  Every lane walks both sides (the wave executes the sum and the product under complementary masks):
// twice the arithmetic per element of the LDS form, no LDS traffic and no barrier
template <int LH>
__device__ __forceinline__ u64 wave_stage(u64 x, int lane, const u64 *tws) {
    const u64 y = lane_xor<LH>(x);
    const int i = lane & ((1 << LH) - 1);
    if (!((lane >> LH) & 1)) return gl_add(x, y);
    const u64 d = gl_sub(y, x);
    return i ? gl_mul(d, tws[(u64)i << (12 - (LH + 1))]) : d;
}

// WAVE: the last min(logn, 6) stages (halves 32 .. 1) run inside a wave on lane exchanges (zp_set_tuning "ntt_small_wave"; the A/B against
// the all-LDS form is profiles/r5_dpp_ab.txt)
template <bool WAVE>
__global__ void __launch_bounds__(256) ntt_small_kernel(SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 lds[];
    const int tid = threadIdx.x;
    const int n = 1 << a.logn;
    const u64 *src = a.in + (u64)blockIdx.x * a.in_cs;
    u64 *dst = a.out + (u64)blockIdx.x * a.out_cs;
    for (int i = tid; i < n; i += 256) lds[i] = (u64)i < a.in_valid ? src[i] : 0ULL;
    __syncthreads();
    const int lds_stages = WAVE ? (a.logn > 6 ? a.logn - 6 : 0) : a.logn;
    for (int s = 0; s < lds_stages; s++) {
        const int lh = a.logn - 1 - s;  // log2(half)
        const int half = 1 << lh;
        for (int b = tid; b < (n >> 1); b += 256) {
            const int i = b & (half - 1);
            const int i0 = ((b >> lh) << (lh + 1)) + i, i1 = i0 + half;
            u64 x = lds[i0], y = lds[i1];
            lds[i0] = gl_add(x, y);
            u64 d = gl_sub(x, y);
            lds[i1] = i ? gl_mul(d, a.tws[(u64)i << (12 - (lh + 1))]) : d;
        }
        __syncthreads();
    }
    if constexpr (WAVE) {
        const int lane = tid & 63;
        for (int base = (tid >> 6) << 6; base < n; base += 256) {       // (n < 64: one partial block; lanes >= n carry zeros nobody stores)
            u64 x = base + lane < n ? lds[base + lane] : 0ULL;
            const int top = a.logn < 6 ? a.logn : 6;                    // stages with half 2^(top-1) .. 1
            if (top >= 6) x = wave_stage<5>(x, lane, a.tws);
            if (top >= 5) x = wave_stage<4>(x, lane, a.tws);
            if (top >= 4) x = wave_stage<3>(x, lane, a.tws);
            if (top >= 3) x = wave_stage<2>(x, lane, a.tws);
            if (top >= 2) x = wave_stage<1>(x, lane, a.tws);
            if (top >= 1) x = wave_stage<0>(x, lane, a.tws);
            const int i = base + lane;
            if (i < n) {
                const int k = brev(i, a.logn);
                if (a.flags & 2) x = gl_mul(x, a.scale);
                if (a.flags & 4) x = gl_mul(x, tw_lookup(a.csl, a.csh, a.cslb, (u64)k));
                dst[k] = x;
            }
        }
        return;
    }
    for (int i = tid; i < n; i += 256) {
        const int k = brev(i, a.logn);
        u64 x = lds[i];
        if (a.flags & 2) x = gl_mul(x, a.scale);
        if (a.flags & 4) x = gl_mul(x, tw_lookup(a.csl, a.csh, a.cslb, (u64)k));
        dst[k] = x;
    }
}

// out[i] = in[i] * shift^i  (coefficients -> coset-scaled coefficients), 2 elements per thread
__global__ void __launch_bounds__(256) coset_scale_kernel(const u64 *in, u64 *out, u64 n, const u64 *lo,
                                                         const u64 *hi, int lb) {
    const u64 col = blockIdx.y;
    const u64 i = ((u64)blockIdx.x * 256 + threadIdx.x);
    if (i < n) out[col * n + i] = gl_mul(in[col * n + i], tw_lookup(lo, hi, lb, i));
}

// four-step NTT of one column split over GPUs: between the two local transforms every element of the
// N2/G x N1 block is multiplied by w_N^(i2 * k1):  rows[r][k] *= w_N^((row0 + r) * k)
__global__ void __launch_bounds__(256) twiddle_rows_kernel(u64 *rows, int logn_row, u64 total, u64 row0, u64 nmask,
                                                           const u64 *lo, const u64 *hi, int lb) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const u64 r = i >> logn_row, k = i & ((1ULL << logn_row) - 1);
    const u64 e = ((row0 + r) * k) & nmask;
    rows[i] = gl_mul(rows[i], tw_lookup(lo, hi, lb, e));
}

// the table of the transposing pass (MODE 3): out[u * R + k] = w^(u k mod N), N = 2^logn, R = 2^L
__global__ void __launch_bounds__(256) tw1_fill_kernel(u64 *out, int logn, int L, const u64 *lo, const u64 *hi, int lb) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >> logn) return;
    const u64 u = i >> L, k = i & ((1ULL << L) - 1);
    out[i] = tw_lookup(lo, hi, lb, (u * k) & ((1ULL << logn) - 1));
}

template <int A1, int A2, int A3, int LOGT, bool BIG = false>
int32_t launch_pass2(zp_ctx *ctx, const PassArgs &a, bool transpose, int W) {
    using G = Geo<A1, A2, A3, LOGT>;
    const u64 tiles = (1ULL << (a.logn - G::L)) >> LOGT;
    // tiles per workgroup: a power of two that divides the tile count and (passes >= 2) keeps a
    // workgroup inside one s-block only by construction of the table per tile (no constraint)
    int tpw = ctx->tune_tpw;
    while (tpw > 1 && (tiles % tpw != 0 || tiles / tpw * (u64)W < (u64)ctx->num_cu * 4)) tpw >>= 1;
    dim3 grid((unsigned)(tiles / tpw), (unsigned)W), block(G::NT);
    // tile + (per-tile table: non-transposing passes with a multiplication) + (LDS copies of the inter-round twiddles, L <= 10)
    const bool has_tab = !transpose && (a.flags & 7) != 0;
    // limb-form butterflies (gl_limb.hpp; knob ntt_limb): every pass with LDS twiddle copies except a first pass without its table
    constexpr bool CAN_LIMB = !BIG && G::L <= 8;      // (radix 2^9 and up: tiles of 64 / 128 KiB leave no room for the 32-byte table records)
    const bool limb = CAN_LIMB && ctx->tune_ntt_limb != 0 && !(transpose && !(A3 == 0 && a.tw1 && (grid.x & 7u) == 0));
    const size_t rec = limb ? 4 : 1;
    const size_t shmem = ((size_t)G::R * G::T + (has_tab ? G::R * rec : 0) + (G::L <= 10 ? (G::R + (1 << (A2 + A3))) * rec : 0)) * sizeof(u64);
    if constexpr (!BIG && A3 == 0) {
        if (transpose && a.tw1 && (grid.x & 7u) == 0) {
            const bool padded = a.in_valid != (1ULL << a.logn);
            auto k = limb ? (padded ? ntt_pass2_kernel<A1, A2, A3, LOGT, true, true, 3, false, CAN_LIMB> : ntt_pass2_kernel<A1, A2, A3, LOGT, true, false, 3, false, CAN_LIMB>)
                          : (padded ? ntt_pass2_kernel<A1, A2, A3, LOGT, true, true, 3, false> : ntt_pass2_kernel<A1, A2, A3, LOGT, true, false, 3, false>);
            PassArgs b = a;
            b.ncols = W;
            hipLaunchKernelGGL(k, dim3(grid.x * (unsigned)W), block, shmem, ctx->stream, b, tpw);
            ZP_HIP(ctx, hipGetLastError());
            return ZP_OK;
        }
    }
    if (transpose) {
        const bool padded = a.in_valid != (1ULL << a.logn);
        auto k = padded ? ntt_pass2_kernel<A1, A2, A3, LOGT, true, true, 1, BIG> : ntt_pass2_kernel<A1, A2, A3, LOGT, true, false, 1, BIG>;
        if (shmem > 65536) ZP_HIP(ctx, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, a, tpw);
    } else {
        auto k = limb ? ((a.flags & 4) ? ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 2, BIG, CAN_LIMB>
                         : (a.flags & 3) ? ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 1, BIG, CAN_LIMB>
                                         : ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 0, BIG, CAN_LIMB>)
                      : ((a.flags & 4) ? ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 2, BIG>
                         : (a.flags & 3) ? ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 1, BIG>
                                         : ntt_pass2_kernel<A1, A2, A3, LOGT, false, false, 0, BIG>);
        if (shmem > 65536) ZP_HIP(ctx, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipLaunchKernelGGL(k, grid, block, shmem, ctx->stream, a, tpw);
    }
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

int32_t dispatch_pass(zp_ctx *ctx, const NttPass &p, const PassArgs &a, bool transpose, int W) {
    if (a.logn > 28) {   // 64-bit lane offsets: only the shapes the default plan uses above 2^28 rows (digits 7 and 8)
        switch (p.L) {
            case 7: return launch_pass2<4, 3, 0, 5, true>(ctx, a, transpose, W);
            case 8: return launch_pass2<4, 4, 0, 4, true>(ctx, a, transpose, W);
            default: ctx->err = "pass radix not built for transforms above 2^28 rows"; return ZP_ERR_UNSUPPORTED;
        }
    }
    switch (p.L) {
        case 5: return launch_pass2<3, 2, 0, 5>(ctx, a, transpose, W);
        case 6: return launch_pass2<3, 3, 0, 5>(ctx, a, transpose, W);
        case 7: return launch_pass2<4, 3, 0, 5>(ctx, a, transpose, W);
        case 8: return ctx->tune_logt == 4 ? launch_pass2<4, 4, 0, 4>(ctx, a, transpose, W) : launch_pass2<4, 4, 0, 5>(ctx, a, transpose, W);
        case 9: return ctx->tune_logt9 == 4 ? launch_pass2<3, 3, 3, 4>(ctx, a, transpose, W) : launch_pass2<3, 3, 3, 5>(ctx, a, transpose, W);
        // 1024-thread workgroups, 128 KiB tiles: two-pass plans up to 2^20 / 2^22 / 2^24 (runs of 128 / 64 / 32 bytes)
        case 10: return launch_pass2<4, 3, 3, 4>(ctx, a, transpose, W);
        case 11: return launch_pass2<4, 4, 3, 3>(ctx, a, transpose, W);
        // (round 6, review item 4: knob ntt_logt12 = 1 -- 512-thread workgroups on 64-KiB tiles of 2 columns, TWO workgroups per CU, 16-byte runs;
        //  a pass with a per-tile twiddle table needs 32 KiB more and stays on the 4-column tile)
        case 12: return (ctx->tune_logt12 == 1 && (transpose || (a.flags & 7) == 0)) ? launch_pass2<4, 4, 4, 1>(ctx, a, transpose, W)
                                                                                     : launch_pass2<4, 4, 4, 2>(ctx, a, transpose, W);
        default: ctx->err = "unsupported pass radix"; return ZP_ERR_UNSUPPORTED;
    }
}

void split_digit(NttPass &p) {
    switch (p.L) {
        case 5: p.A1 = 3; p.A2 = 2; p.A3 = 0; break;
        case 6: p.A1 = 3; p.A2 = 3; p.A3 = 0; break;
        case 7: p.A1 = 4; p.A2 = 3; p.A3 = 0; break;
        case 8: p.A1 = 4; p.A2 = 4; p.A3 = 0; break;
        case 10: p.A1 = 4; p.A2 = 3; p.A3 = 3; break;
        case 11: p.A1 = 4; p.A2 = 4; p.A3 = 3; break;
        case 12: p.A1 = 4; p.A2 = 4; p.A3 = 4; break;
        default: p.A1 = 3; p.A2 = 3; p.A3 = 3; break;
    }
    p.logT = 5;
}

}  // namespace

int32_t zpi_scratch(zp_ctx *ctx, int which, size_t elems, u64 **out) {
    if (ctx->scratch_elems[which] < elems) {
        if (ctx->scratch[which]) {
            ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ZP_HIP(ctx, hipFree(ctx->scratch[which]));
            ctx->scratch[which] = nullptr;
            ctx->scratch_elems[which] = 0;
        }
        ZP_HIP(ctx, hipMalloc((void **)&ctx->scratch[which], elems * sizeof(u64)));
        ctx->scratch_elems[which] = elems;
    }
    *out = ctx->scratch[which];
    return ZP_OK;
}

int32_t zpi_pinned(zp_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_bytes < bytes) {
        if (ctx->pinned) {
            ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ZP_HIP(ctx, hipHostFree(ctx->pinned));
            ctx->pinned = nullptr;
            ctx->pinned_bytes = 0;
        }
        size_t cap = bytes < (8u << 20) ? (8u << 20) : bytes;
        ZP_HIP(ctx, hipHostMalloc(&ctx->pinned, cap, hipHostMallocMapped | hipHostMallocPortable));
        ctx->pinned_bytes = cap;
    }
    *out = ctx->pinned;
    return ZP_OK;
}

// Small host<->device copies are KERNELS through the pinned (device-visible) staging buffer, not DMA copies: a
// hipMemcpyAsync of a few hundred bytes queues behind whatever the copy engines are doing -- with a second ctx
// streaming 512 MB witnesses in, the transcript's tiny copies waited for all of them (0.47 s in the first chunk
// of a 16-chunk batch).  A copy kernel is ordered on the ctx stream with the compute it feeds and never meets
// the DMA queues.
__global__ void __launch_bounds__(256) small_copy_kernel(const u32 *__restrict__ src, u32 *__restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

static inline bool word_copyable(const void *a, size_t bytes) { return (((uintptr_t)a | bytes) & 3u) == 0; }

static inline void launch_small_copy(zp_ctx *ctx, const void *src, void *dst, size_t bytes) {
    const size_t n = bytes / 4;
    const unsigned blocks = (unsigned)(n < 256 * 256 ? (n + 255) / 256 : 256);
    hipLaunchKernelGGL(small_copy_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, (const u32 *)src, (u32 *)dst, n);
}

int32_t zpi_d2h_small(zp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    void *st;
    ZP_TRY(zpi_pinned(ctx, bytes, &st));
    if (word_copyable(d_src, bytes)) {
        launch_small_copy(ctx, d_src, st, bytes);
        ZP_HIP(ctx, hipGetLastError());
    } else {
        ZP_HIP(ctx, hipMemcpyAsync(st, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(h_dst, st, bytes);
    return ZP_OK;
}

int32_t zpi_h2d_small(zp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    void *st;
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the staging buffer may still feed an earlier copy
    ZP_TRY(zpi_pinned(ctx, bytes, &st));
    memcpy(st, h_src, bytes);
    if (word_copyable(d_dst, bytes)) {
        launch_small_copy(ctx, st, d_dst, bytes);
        ZP_HIP(ctx, hipGetLastError());
    } else {
        ZP_HIP(ctx, hipMemcpyAsync(d_dst, st, bytes, hipMemcpyHostToDevice, ctx->stream));
    }
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

static int32_t upload(zp_ctx *ctx, const std::vector<u64> &h, u64 **d) {
    ZP_HIP(ctx, hipMalloc((void **)d, h.size() * sizeof(u64)));
    ZP_HIP(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(u64), hipMemcpyHostToDevice));
    return ZP_OK;
}

// role (the two sides of an extension's seam, zpi_lde): 0 = the default plan; 1 = a plan that ENDS in a radix-256 pass (the inverse
// transform in front of lde_seam_kernel); 2 = one that STARTS with a radix-256 pass (the forward transform behind it).  The digit 8 is
// fixed, the other logn - 8 bits are split over as few passes as the default plan would use for them, the first of them >= 7 where that
// leaves >= 5 for the rest (a first pass of radix >= 2^7 takes its twiddles from the shared table).  ZP_ERR_UNSUPPORTED when that takes
// more passes than the default plan (the caller then runs the two transforms unfused).
int32_t zpi_get_plan_role(zp_ctx *ctx, int logn, bool inverse, int role, NttPlan **out) {
    const int maxl = (ctx->tune_ntt_maxl >= 6 && ctx->tune_ntt_maxl <= 12) ? ctx->tune_ntt_maxl : 9;
    const int order = ctx->tune_ntt_order;          // 0: auto, 1: larger digits first, 2: larger digits last
    const int key = (((logn * 2 + (inverse ? 1 : 0)) * 16 + maxl) * 4 + (order & 3)) * 4 + (role & 3);
    int rdig[6], nr = 0;
    if (role) {
        const int r = logn - 8, m_default = (logn + maxl - 1) / maxl;
        if (logn <= 12 || r < 5 || maxl != 9) return ZP_ERR_UNSUPPORTED;
        nr = (r + maxl - 1) / maxl;
        if (1 + nr > m_default || nr > 4) return ZP_ERR_UNSUPPORTED;
        int rem = r;
        for (int i = 0; i < nr; i++) {
            int d = (rem + (nr - i) - 1) / (nr - i);                        // balanced, larger first
            if (i == 0 && nr >= 2 && d < 7 && rem - 7 >= 5 * (nr - 1)) d = 7;
            rdig[i] = d;
            rem -= d;
        }
        for (int i = 0; i < nr; i++)
            if (rdig[i] < 5 || rdig[i] > 9) return ZP_ERR_UNSUPPORTED;
    }
    auto it = ctx->plans.find(key);
    if (it != ctx->plans.end()) {
        *out = &it->second;
        return ZP_OK;
    }
    NttPlan pl;
    pl.logn = logn;
    pl.inverse = inverse;
    u64 w = gl_root(ctx->root32, logn);
    u64 w4096 = gl_root(ctx->root32, 12);
    u64 w16 = gl_root(ctx->root32, 4);
    if (inverse) {
        w = gl_inv(w);
        w4096 = gl_inv(w4096);
        w16 = gl_inv(w16);
    }
    pl.ninv = gl_inv((1ULL << logn) % GL_P);
    pl.j0inv = 0;
    for (int j = 1; j < 16; j += 2)
        if (gl_pow(1ULL << 12, (u64)j) == w16)
            for (int ji = 1; ji < 16; ji += 2)
                if (((j * ji) & 15) == 1) pl.j0inv = ji;
    pl.w16[0] = 1;
    for (int i = 1; i < 8; i++) pl.w16[i] = gl_mul(pl.w16[i - 1], w16);
    std::vector<u64> tws(4096);
    tws[0] = 1;
    for (int i = 1; i < 4096; i++) tws[i] = gl_mul(tws[i - 1], w4096);
    ZP_TRY(upload(ctx, tws, &pl.d_tws));
    if (logn > 12 && role) {
        int logP = 0;
        pl.npass = nr + 1;
        for (int i = 0; i < pl.npass; i++) {
            NttPass &p = pl.pass[i];
            p.L = role == 1 ? (i < nr ? rdig[i] : 8) : (i == 0 ? 8 : rdig[i - 1]);
            split_digit(p);
            p.logPprev = logP;
            logP += p.L;
        }
    } else if (logn > 12) {
        const int m = (logn + maxl - 1) / maxl;
        int rem = logn;
        int logP = 0;
        pl.npass = m;
        int digits[6];
        for (int i = 0; i < m; i++) {
            digits[i] = (rem + (m - i) - 1) / (m - i);  // balanced, larger digits first
            rem -= digits[i];
        }
        for (int i = 0; i < m; i++) {
            NttPass &p = pl.pass[i];
            // the transposing first pass pays a per-lane twiddle chain on top of its rounds: a three-round radix-512 digit is
            // cheaper as the (plain) last pass; with digits <= 8 the larger-first order measured 1-4 % faster
            // (profiles/r2_order_sweep.txt)
            const bool last = order == 2 || (order == 0 && digits[0] == 9 && digits[m - 1] < 9);
            p.L = last ? digits[m - 1 - i] : digits[i];
            split_digit(p);
            p.logPprev = logP;
            logP += p.L;
        }
    }
    {   // two-level table of w^e, e < N (also used by the FRI fold for w_n^-i)
        pl.lb = (logn + 1) / 2;
        std::vector<u64> lo(1ULL << pl.lb), hi(1ULL << (logn - pl.lb));
        lo[0] = 1;
        for (size_t i = 1; i < lo.size(); i++) lo[i] = gl_mul(lo[i - 1], w);
        u64 wl = gl_mul(lo.back(), w);  // w^(2^lb)
        hi[0] = 1;
        for (size_t i = 1; i < hi.size(); i++) hi[i] = gl_mul(hi[i - 1], wl);
        ZP_TRY(upload(ctx, lo, &pl.d_twl));
        ZP_TRY(upload(ctx, hi, &pl.d_twh));
    }
    auto ins = ctx->plans.emplace(key, pl);
    *out = &ins.first->second;
    return ZP_OK;
}

int32_t zpi_get_plan(zp_ctx *ctx, int logn, bool inverse, NttPlan **out) { return zpi_get_plan_role(ctx, logn, inverse, 0, out); }

// the two plans of a fused extension (blow-up 2) of 2^logn-row columns: the default plans where they already meet in radix-256 passes,
// else the seam-role plans (knob lde_seam_plans: 0 = default plans only).  false: no fused path at this size.
static bool seam_plans(zp_ctx *ctx, int logn, NttPlan **pi, NttPlan **pf) {
    auto fits = [](const NttPlan *i, const NttPlan *f) {
        const NttPass &li = i->pass[i->npass - 1], &ff = f->pass[0];
        return i->npass >= 2 && f->npass >= 2 && li.L == 8 && li.A1 == 4 && li.A2 == 4 && li.A3 == 0 && ff.L == 8 && ff.A1 == 4 && ff.A2 == 4 && ff.A3 == 0;
    };
    NttPlan *di = nullptr, *df = nullptr;
    if (zpi_get_plan(ctx, logn, true, &di) != ZP_OK || zpi_get_plan(ctx, logn + 1, false, &df) != ZP_OK) return false;
    NttPlan *ci = di, *cf = df;
    // measured on one box (profiles/r5_lde_seam_plans_ab.txt): the seam plans win 1-3 % from 2^21 rows on ((7,6,8)+(8,7,7) .. (8,7,8)+(8,8,8)) and LOSE
    // 5-11 % at 2^19 / 2^20, where fixing one digit at 8 leaves radix-32 passes ((6,5,8)+(8,7,5), (7,5,8)+(8,7,6)): default from 2^21 (knob 2: always)
    if (!fits(ci, cf) && (ctx->tune_lde_seam_plans == 2 || (ctx->tune_lde_seam_plans == 1 && logn >= 21))) {
        const NttPass &li = di->pass[di->npass - 1], &ff = df->pass[0];
        if (!(di->npass >= 2 && li.L == 8 && li.A3 == 0) && zpi_get_plan_role(ctx, logn, true, 1, &ci) != ZP_OK) return false;
        if (!(df->npass >= 2 && ff.L == 8 && ff.A3 == 0) && zpi_get_plan_role(ctx, logn + 1, false, 2, &cf) != ZP_OK) return false;
    }
    if (!fits(ci, cf)) return false;
    *pi = ci;
    *pf = cf;
    return true;
}

int32_t zpi_get_coset(zp_ctx *ctx, int logn, u64 shift, u64 pre, CosetTable **out) {
    for (auto &c : ctx->cosets)
        if (c.logn == logn && c.shift == shift && c.pre == pre) {
            *out = &c;
            return ZP_OK;
        }
    CosetTable c;
    c.logn = logn;
    c.shift = shift;
    c.pre = pre;
    c.lb = (logn + 1) / 2;
    std::vector<u64> lo(1ULL << c.lb), hi(1ULL << (logn - c.lb));
    lo[0] = pre;
    u64 sp = 1;
    for (size_t i = 1; i < lo.size(); i++) {
        sp = gl_mul(sp, shift);
        lo[i] = gl_mul(pre, sp);
    }
    u64 sl = gl_mul(sp, shift);  // shift^(2^lb)
    hi[0] = 1;
    for (size_t i = 1; i < hi.size(); i++) hi[i] = gl_mul(hi[i - 1], sl);
    ZP_TRY(upload(ctx, lo, &c.d_lo));
    ZP_TRY(upload(ctx, hi, &c.d_hi));
    ctx->cosets.push_back(c);
    *out = &ctx->cosets.back();
    return ZP_OK;
}

// pass i of a plan over w columns: cur (column stride cur_cs, in_valid real elements per column) -> nxt (column stride N)
static int32_t run_one_pass(zp_ctx *ctx, NttPlan *pl, int i, bool inverse, const NttRunOpts &opts, const u64 *cur, u64 cur_cs, u64 in_valid, u64 *nxt, int w) {
    const int logn = pl->logn, m = pl->npass;
    const bool last = (i == m - 1);
    PassArgs a;
    memset(&a, 0, sizeof(a));
    a.in = cur;
    a.out = nxt;
    a.in_cs = cur_cs;
    a.out_cs = 1ULL << logn;
    a.in_valid = in_valid;
    a.twl = pl->d_twl;
    a.twh = pl->d_twh;
    a.lb = pl->lb;
    a.tws = pl->d_tws;
    a.tw1 = (i == 0 && logn <= ctx->tune_ntt_tw1) ? pl->d_tw1 : nullptr;
    a.scale = pl->ninv;
    a.logn = logn;
    a.logPprev = pl->pass[i].logPprev;
    a.j0inv = pl->j0inv;
    a.flags = last ? 0 : 1;
    if (last && inverse) a.flags |= 2;
    if (last && opts.post_scale) {
        a.flags |= 4;
        a.csl = opts.post_scale->d_lo;
        a.csh = opts.post_scale->d_hi;
        a.cslb = opts.post_scale->lb;
    }
    zp_ctx::PassEv ev;
    if (ctx->profiling) {
        ZP_HIP(ctx, hipEventCreate(&ev.a));
        ZP_HIP(ctx, hipEventCreate(&ev.b));
        ev.radix_log = (i == 0) ? -pl->pass[i].L : pl->pass[i].L;
        ZP_HIP(ctx, hipEventRecord(ev.a, ctx->stream));
    }
    ZP_TRY(dispatch_pass(ctx, pl->pass[i], a, i == 0, w));
    if (ctx->profiling) {
        ZP_HIP(ctx, hipEventRecord(ev.b, ctx->stream));
        ctx->pass_events.push_back(ev);
    }
    return ZP_OK;
}
// the first pass's full table (8 bytes per element of ONE column, shared by all columns and all later calls of this size):
// built at the first use of a plan by radix-2^7 / 2^8 two-round passes below 2^29 rows
static int32_t ensure_tw1(zp_ctx *ctx, NttPlan *pl) {
    const int logn = pl->logn;
    if (!pl->d_tw1 && !pl->tw1_unavailable && logn > 12 && logn <= ctx->tune_ntt_tw1 && logn <= 28 && pl->pass[0].A3 == 0 && pl->pass[0].L >= 7) {
        if (hipMalloc((void **)&pl->d_tw1, sizeof(u64) << logn) != hipSuccess) {
            // the table is an optimisation (8 bytes per row, per plan, per ctx): without it the first pass multiplies by per-lane
            // twiddle chains (MODE 1), same results.  Clear the sticky error and do not try again for this plan.
            (void)hipGetLastError();
            pl->d_tw1 = nullptr;
            pl->tw1_unavailable = true;
        } else {
            hipLaunchKernelGGL(tw1_fill_kernel, dim3((unsigned)(((1ULL << logn) + 255) / 256)), dim3(256), 0, ctx->stream, pl->d_tw1, logn, pl->pass[0].L,
                               pl->d_twl, pl->d_twh, pl->lb);
            ZP_HIP(ctx, hipGetLastError());
        }
    }
    return ZP_OK;
}

int32_t zpi_ntt_run(zp_ctx *ctx, const u64 *d_in, u64 *d_out, int logn, int W, bool inverse,
                    const NttRunOpts &opts) {
    ZP_ARG(ctx, logn >= 0 && logn <= 32, "logn must be in [0,32]");
    ZP_ARG(ctx, W >= 0, "W must be >= 0");
    ZP_ARG(ctx, d_in && d_out, "null device pointer");
    if (W == 0) return ZP_OK;
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logn, inverse, &pl));
    const u64 N = 1ULL << logn;
    const u64 in_valid = opts.in_valid_log >= 0 ? (1ULL << opts.in_valid_log) : N;
    ZP_ARG(ctx, in_valid <= N, "in_valid_log > logn");
    ZP_ARG(ctx, !(in_valid != N && (const u64 *)d_out == d_in), "zero-padded input cannot be in place");

    if (logn <= 12) {
        SmallArgs a;
        memset(&a, 0, sizeof(a));
        a.in = d_in;
        a.out = d_out;
        a.in_cs = in_valid;
        a.out_cs = N;
        a.in_valid = in_valid;
        a.tws = pl->d_tws;
        a.scale = pl->ninv;
        a.logn = logn;
        a.flags = inverse ? 2 : 0;
        if (opts.post_scale) {
            a.flags |= 4;
            a.csl = opts.post_scale->d_lo;
            a.csh = opts.post_scale->d_hi;
            a.cslb = opts.post_scale->lb;
        }
        // in place is safe: a block owns its column and loads all of it into LDS before storing
        // in-wave stages where they measured faster (profiles/r5_dpp_ab.txt: <= 64 points 1.1-1.75x; from 2^8 points the doubled arithmetic loses)
        if (ctx->tune_ntt_small_wave == 1 || (ctx->tune_ntt_small_wave == 0 && logn <= 6))
            hipLaunchKernelGGL(ntt_small_kernel<true>, dim3((unsigned)W), dim3(256), (size_t)N * sizeof(u64), ctx->stream, a);
        else
            hipLaunchKernelGGL(ntt_small_kernel<false>, dim3((unsigned)W), dim3(256), (size_t)N * sizeof(u64), ctx->stream, a);
        ZP_HIP(ctx, hipGetLastError());
        return ZP_OK;
    }

    // chunk the columns so that each ping-pong scratch buffer stays <= 2 GiB.  2^28 elements per launch since round 4 (profiles/r4_chunk_sweep.txt:
    // NTT 2^20 .. 2^25 0-3 % faster than at 2^27, the first pass's shared twiddle table now serving 16 columns of a tile; 2^29 adds nothing.
    // Round 2, before that table, measured the opposite: profiles/r2_chunk_sweep.txt)
    const u64 cap_elems = 1ULL << (ctx->tune_ntt_chunk_log > 0 ? ctx->tune_ntt_chunk_log : 28);
    int wc = (int)(cap_elems >> logn);
    if (wc < 1) wc = 1;
    if (wc > W) wc = W;
    const int m = pl->npass;
    ZP_TRY(ensure_tw1(ctx, pl));
    u64 *s0 = nullptr, *s1 = nullptr;
    if (m >= 2) ZP_TRY(zpi_scratch(ctx, 0, (size_t)wc << logn, &s0));
    if (m >= 3) ZP_TRY(zpi_scratch(ctx, 1, (size_t)wc << logn, &s1));

    for (int c0 = 0; c0 < W; c0 += wc) {
        const int w = (W - c0 < wc) ? (W - c0) : wc;
        const u64 *cur = d_in + (u64)c0 * in_valid;
        u64 cur_cs = in_valid;
        for (int i = 0; i < m; i++) {
            const bool last = (i == m - 1);
            u64 *nxt = last ? d_out + ((u64)c0 << logn) : ((i & 1) ? s1 : s0);
            ZP_TRY(run_one_pass(ctx, pl, i, inverse, opts, cur, cur_cs, (i == 0) ? in_valid : N, nxt, w));
            cur = nxt;
            cur_cs = N;
        }
    }
    return ZP_OK;
}

int32_t zpi_lde(zp_ctx *ctx, const u64 *d_in, u64 *d_out, u64 *d_coef, int logn, int logb, int W, u64 shift) {
    ZP_ARG(ctx, logn >= 0 && logb >= 0 && logn + logb <= 32, "logn/logb out of range");
    ZP_ARG(ctx, W >= 0, "W must be >= 0");
    ZP_ARG(ctx, d_in && d_out, "null device pointer");
    ZP_ARG(ctx, d_out != d_in, "zp_lde cannot run in place");
    if (W == 0) return ZP_OK;
    if (shift == 0) shift = ctx->coset_shift;
    ZP_ARG(ctx, shift < GL_P, "shift not canonical");
    const u64 N = 1ULL << logn;
    CosetTable *ct;
    ZP_TRY(zpi_get_coset(ctx, logn, shift, 1, &ct));
    // columns go in chunks so that the scaled-coefficient buffer stays <= 2 GiB
    int wc = (int)((1ULL << (ctx->tune_ntt_chunk_log > 0 ? ctx->tune_ntt_chunk_log : 28)) >> logn);
    if (wc < 1) wc = 1;
    if (wc > W) wc = W;
    // ---- blow-up 2 with matching radix-256 passes on both sides of the seam: the inverse transform's last pass and the forward
    // transform's first pass run as ONE kernel (lde_seam_kernel) and the scaled coefficients never travel (knob lde_seam)
    // Measured (profiles/r4_lde_seam_ab.txt, 2^24 x 32): 19.5 -> 18.7 ms without the coefficient store -- the fused kernel does three tile
    // bodies of integer work for three units of traffic and is bound by the former (0.70 ms against 0.73 ms for the two launches it replaces,
    // and the forward transform's remaining passes run on fewer launches): a pass of this library is BALANCED between its integer work and its
    // traffic, so taking away either alone buys little.  On by default where the caller does not want the coefficients (knob lde_seam: 0 never,
    // 1 default, 2 always).
    if (logb == 1 && (ctx->tune_lde_seam == 2 || (ctx->tune_lde_seam == 1 && !d_coef)) && ctx->tune_logt == 4 && logn >= 16 && logn + 1 <= 28 && logn + 1 <= ctx->tune_ntt_tw1) {
        NttPlan *pi = nullptr, *pf = nullptr;
        const bool have = seam_plans(ctx, logn, &pi, &pf);
        if (have) {
            ZP_TRY(ensure_tw1(ctx, pf));
            ZP_TRY(ensure_tw1(ctx, pi));
        }
        const int tpw = (ctx->tune_seam_tpw == 1 || ctx->tune_seam_tpw == 4) ? ctx->tune_seam_tpw : 2;     // a power of two: divides the tile count
        if (have && pf->d_tw1 && (((N >> 12) / tpw) & 7u) == 0) {
            const int mi = pi->npass, mf = pf->npass;
            const int wf = wc >= 2 ? wc / 2 : 1;                      // columns per forward sub-chunk (2N rows each)
            size_t need = (size_t)wc << logn;
            if (((size_t)wf << (logn + 1)) > need) need = (size_t)wf << (logn + 1);
            u64 *s0 = nullptr, *s1 = nullptr, *s2 = nullptr;
            ZP_TRY(zpi_scratch(ctx, 0, need, &s0));
            ZP_TRY(zpi_scratch(ctx, 1, need, &s1));
            if (mf >= 3) ZP_TRY(zpi_scratch(ctx, 2, (size_t)wf << (logn + 1), &s2));
            NttRunOpts none;
            for (int c0 = 0; c0 < W; c0 += wc) {
                const int w = (W - c0 < wc) ? (W - c0) : wc;
                const u64 *cur = d_in + (u64)c0 * N;
                for (int i = 0; i + 1 < mi; i++) {                     // the inverse transform up to its last pass
                    u64 *nxt = (i & 1) ? s1 : s0;
                    ZP_TRY(run_one_pass(ctx, pi, i, true, none, cur, N, N, nxt, w));
                    cur = nxt;
                }
                u64 *seam_out = (cur == s0) ? s1 : s0;
                for (int h0 = 0; h0 < w; h0 += wf) {
                    const int wh = (w - h0 < wf) ? (w - h0) : wf;
                    SeamArgs sa;
                    memset(&sa, 0, sizeof(sa));
                    sa.in = cur + (u64)h0 * N;
                    sa.out = seam_out;
                    sa.coef = d_coef ? d_coef + (u64)(c0 + h0) * N : nullptr;
                    sa.tws_i = pi->d_tws;
                    sa.tws_f = pf->d_tws;
                    sa.tw1 = pf->d_tw1;
                    sa.csl = ct->d_lo;
                    sa.csh = ct->d_hi;
                    sa.cslb = ct->lb;
                    sa.scale = pi->ninv;
                    sa.logn = logn;
                    sa.j0inv_i = pi->j0inv;
                    sa.j0inv_f = pf->j0inv;
                    sa.ncols = wh;
                    zp_ctx::PassEv ev;
                    if (ctx->profiling) {
                        ZP_HIP(ctx, hipEventCreate(&ev.a));
                        ZP_HIP(ctx, hipEventCreate(&ev.b));
                        ev.radix_log = 88;                              // the seam: inverse radix 2^8 + forward radix 2^8
                        ZP_HIP(ctx, hipEventRecord(ev.a, ctx->stream));
                    }
                    hipLaunchKernelGGL(lde_seam_kernel, dim3((unsigned)(((N >> 12) / tpw) * (u64)wh)), dim3(256), (size_t)(256 * 16 + 3 * 256) * sizeof(u64), ctx->stream, sa, tpw);
                    ZP_HIP(ctx, hipGetLastError());
                    if (ctx->profiling) {
                        ZP_HIP(ctx, hipEventRecord(ev.b, ctx->stream));
                        ctx->pass_events.push_back(ev);
                    }
                    const u64 *fc = seam_out;                           // the forward transform from its second pass on
                    u64 *out = d_out + ((u64)(c0 + h0) << (logn + 1));
                    for (int i = 1; i < mf; i++) {
                        u64 *nxt = (i == mf - 1) ? out : s2;
                        ZP_TRY(run_one_pass(ctx, pf, i, false, none, fc, 2 * N, 2 * N, nxt, wh));
                        fc = nxt;
                    }
                }
            }
            return ZP_OK;
        }
    }
    u64 *scaled = nullptr;
    if (!d_coef) ZP_TRY(zpi_scratch(ctx, 2, (size_t)wc << logn, &scaled));
    for (int c0 = 0; c0 < W; c0 += wc) {
        const int w = (W - c0 < wc) ? (W - c0) : wc;
        const u64 *in = d_in + (u64)c0 * N;
        u64 *out = d_out + ((u64)c0 << (logn + logb));
        NttRunOpts inv;
        inv.post_scale = ct;  // the inverse transform's last pass multiplies c_i by shift^i
        // d_coef (optional) IS the scaled-coefficient buffer: the interpolant composed with the coset shift, c_i * shift^i.
        // Nothing downstream needs the plain c_i: p(z) = sum_i (c_i shift^i) (z / shift)^i  (stark/prover.py evaluates there),
        // so the separate copy + scaling pass of round 1 (32*N bytes per column) is gone.
        u64 *sc = d_coef ? d_coef + (u64)c0 * N : scaled;
        ZP_TRY(zpi_ntt_run(ctx, in, sc, logn, w, true, inv));
        NttRunOpts fwd;
        fwd.in_valid_log = logn;  // zero padding is implicit: rows >= N read as 0
        ZP_TRY(zpi_ntt_run(ctx, sc, out, logn + logb, w, false, fwd));
    }
    return ZP_OK;
}

// what zp_lde(blow-up 2) does at this size, as JSON (zp_ntt_plan_json's "lde" member): whether the seam kernel is taken and the radices
// on both sides of it
int32_t zpi_lde_plan_json(zp_ctx *ctx, int logn, int want_coef, std::string *out) {
    NttPlan *pi = nullptr, *pf = nullptr;
    const int tpw = (ctx->tune_seam_tpw == 1 || ctx->tune_seam_tpw == 4) ? ctx->tune_seam_tpw : 2;
    bool seam = (ctx->tune_lde_seam == 2 || (ctx->tune_lde_seam == 1 && !want_coef)) && ctx->tune_logt == 4 && logn >= 16 && logn + 1 <= 28 &&
                logn + 1 <= ctx->tune_ntt_tw1 && seam_plans(ctx, logn, &pi, &pf) && ((((1ULL << logn) >> 12) / tpw) & 7u) == 0;
    if (!seam) {
        if (logn + 1 > 32) return ZP_ERR_ARG;
        ZP_TRY(zpi_get_plan(ctx, logn, true, &pi));
        ZP_TRY(zpi_get_plan(ctx, logn + 1, false, &pf));
    }
    auto digits = [](const NttPlan *p) {
        std::string d = "[";
        for (int i = 0; i < p->npass; i++) d += (i ? ", " : "") + std::to_string(p->pass[i].L);
        return d + "]";
    };
    *out = std::string("{\"blowup\": 2, \"coefficients_stored\": ") + (want_coef ? "true" : "false") + ", \"seam_fused\": " + (seam ? "true" : "false") +
           ", \"inverse_radix_logs\": " + digits(pi) + ", \"forward_radix_logs\": " + digits(pf) + "}";
    return ZP_OK;
}

int32_t zpi_twiddle_rows(zp_ctx *ctx, u64 *d_rows, int logn_row, int W, u64 row0, int logn_total, bool inverse) {
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logn_total, inverse, &pl));
    const u64 total = (u64)W << logn_row;
    const u64 nmask = logn_total >= 64 ? ~0ULL : ((1ULL << logn_total) - 1);
    hipLaunchKernelGGL(twiddle_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, d_rows, logn_row,
                       total, row0, nmask, pl->d_twl, pl->d_twh, pl->lb);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

// ---- measurement: what this device sustains on a plain copy, with this library's own kernel (16 bytes per lane, one persistent workgroup
// per CU by default since round 5) -- the ceiling bench.py prints next to the vendor peak
typedef __attribute__((ext_vector_type(4))) unsigned int zp_u32x4;
// U loads of 16 bytes in flight per lane, then U stores; NT: non-temporal policy (the data is touched once) or the default one.  Which grid,
// block size, depth and policy stream fastest is MEASURED (tools/ntt_r5_ab.py -> profiles/r5_ntt_ab.txt; knobs copy_grid / copy_block /
// copy_unroll / copy_nt): round 5 found one workgroup per CU (256 x 256 lanes, 4 in flight) at 5.6 TB/s against 4.9 for the 2048-workgroup
// grid of rounds 1-4 -- fewer concurrent streams, not more, is what HBM3E wants.
template <int U, bool NT>
__global__ void __launch_bounds__(1024) hbm_copy_kernel(const zp_u32x4 *__restrict__ in, zp_u32x4 *__restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        zp_u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(in + i + k * stride) : in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) {
            if (NT) __builtin_nontemporal_store(v[k], out + i + k * stride);
            else out[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) out[i] = in[i];
}

extern "C" int32_t zp_hbm_copy_probe(zp_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int32_t reps, float *ms_per_copy) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, d_src && d_dst && ms_per_copy && reps >= 1 && (bytes & 15) == 0 && bytes >= 16, "bad arguments");
    hipEvent_t e0, e1;
    ZP_HIP(ctx, hipEventCreate(&e0));
    ZP_HIP(ctx, hipEventCreate(&e1));
    const unsigned grid = ctx->tune_copy_grid > 0 ? (unsigned)ctx->tune_copy_grid : 256u;
    const unsigned block = (ctx->tune_copy_block == 512 || ctx->tune_copy_block == 1024) ? (unsigned)ctx->tune_copy_block : 256u;
    const bool u8 = ctx->tune_copy_unroll == 8;
    auto k = ctx->tune_copy_nt ? (u8 ? hbm_copy_kernel<8, true> : hbm_copy_kernel<4, true>) : (u8 ? hbm_copy_kernel<8, false> : hbm_copy_kernel<4, false>);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, ctx->stream, (const zp_u32x4 *)d_src, (zp_u32x4 *)d_dst, bytes / 16);
    ZP_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, ctx->stream, (const zp_u32x4 *)d_src, (zp_u32x4 *)d_dst, bytes / 16);
    ZP_HIP(ctx, hipEventRecord(e1, ctx->stream));
    ZP_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0.f;
    ZP_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *ms_per_copy = ms / reps;
    return ZP_OK;
}

// ---- layout kernels of the multi-GPU paths: the send-buffer packing of the column->row all-to-all and the
// local transposes of the four-step NTT.  (Round 1 did these with generic tensor copies: 21 G elements/s for a four-step
// transform on one GPU against 75 G for the plain one.)
// out[h][w][j] = in[w][h*Mg + j]:  [Wl][G*Mg] -> [G][Wl][Mg]; runs of Mg contiguous elements, 16 bytes per lane
__global__ void __launch_bounds__(256) pack_blocks_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 Wl, u64 G, u64 Mg) {
    const u64 total2 = Wl * G * Mg / 2;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < total2; i += (u64)gridDim.x * 256) {
        const u64 e = 2 * i, j = e % Mg, w = (e / Mg) % Wl, h = e / (Mg * Wl);
        const u64 *src = in + w * (G * Mg) + h * Mg + j;
        u64 *dst = out + e;
        dst[0] = ZP_LDG(src);
        dst[1] = ZP_LDG(src + 1);
    }
}
// out[c][r] = in[r][c], 64 x 64 tiles through LDS (row length 65: the column reads of the write phase hit 64 banks)
__global__ void __launch_bounds__(256) transpose_u64_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, u64 R, u64 C) {
    __shared__ u64 tile[64][65];
    const u64 tiles_c = (C + 63) / 64;
    const u64 r0 = (blockIdx.x / tiles_c) * 64, c0 = (blockIdx.x % tiles_c) * 64;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 r = r0 + i * 4 + w, c = c0 + lane;
        if (r < R && c < C) tile[i * 4 + w][lane] = ZP_LDG(&in[r * C + c]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const u64 c = c0 + i * 4 + w, r = r0 + lane;
        if (r < R && c < C) ZP_STG(&out[c * R + r], tile[lane][i * 4 + w]);
    }
}

extern "C" int32_t zp_pack_blocks(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t rows, size_t row_len, int32_t parts) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "pack_blocks");
    ZP_ARG(ctx, d_in && d_out && d_in != d_out && parts >= 1 && row_len % (size_t)parts == 0, "bad arguments");
    const size_t Mg = row_len / parts;
    ZP_ARG(ctx, Mg % 2 == 0 || rows * row_len == 0, "part length must be even");
    if (rows * row_len == 0) return ZP_OK;
    const size_t total2 = rows * row_len / 2;
    const unsigned blocks = (unsigned)(total2 / 256 + 1 < 8192 ? total2 / 256 + 1 : 8192);
    hipLaunchKernelGGL(pack_blocks_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const u64 *)d_in, (u64 *)d_out, (u64)rows, (u64)parts, (u64)Mg);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

extern "C" int32_t zp_transpose(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t rows, size_t cols) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "transpose");
    ZP_ARG(ctx, d_in && d_out && d_in != d_out, "bad arguments");
    if (rows * cols == 0) return ZP_OK;
    const size_t tiles = ((rows + 63) / 64) * ((cols + 63) / 64);
    ZP_ARG(ctx, tiles < (1ULL << 31), "matrix too large for one launch");
    hipLaunchKernelGGL(transpose_u64_kernel, dim3((unsigned)tiles), dim3(256), 0, ctx->stream, (const u64 *)d_in, (u64 *)d_out, (u64)rows, (u64)cols);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
