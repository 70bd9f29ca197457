// The dense linear layer of Poseidon-BN254 (t = 17) on the matrix cores, gfx950 (measurement tool).
//
// out_r = sum_c M[r][c] * s_c / R mod r  for r = 0..16: seventeen 17-term dot products of 254-bit values with CONSTANT M -- the one shape in this
// repository where a limb-product formulation has a shared operand over many independent products (the single-product form,
// tools/ubench_mfma_fq.hip, lost to the VALU because every product paid a 64-instruction recombination; here ONE recombination closes a
// 17-term sum).  Two ways, same inputs, results compared word for word:
//   (a) VALU, the production form (csrc/poseidon_bn254.hip: fr_dotc): one lane per state, 9 x 29-bit limbs, 17 x 81 v_mad_u64_u32 + three
//       Montgomery reductions per output.
//   (b) MFMA: one wave = 32 states, two lanes per state (lane l: state l & 31, half h = l >> 5).  Values as 32 balanced base-256 digits
//       (V + 0x80..80 with carries, XOR 0x80..80: 16 instructions); A = Toeplitz blocks of M's digits (two 32 x 32 row blocks per constant), B =
//       the digits of s_c: out_r's 63 signed column sums = 2 x 17 v_mfma_i32_32x32x32_i8; each half-lane folds its 32 column sums into 29-bit
//       positions (one v_mad_i64_i32 each), the halves of an output pair meet by v_permlane32_swap, one signed Montgomery reduction.
//       An element is owned by the half c & 1 of its state's lane pair (owner: S-box, digit conversion); digits reach the B layout by four swaps
//       per element pair.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I eigen_zeth_amd/csrc -o tools/ubench_mfma_mds tools/ubench_mfma_mds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fr254.hpp"
typedef long long i64;
typedef int i32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef i32 i32x16 __attribute__((ext_vector_type(16)));
typedef i32 i32x4 __attribute__((ext_vector_type(4)));
constexpr int T = 17;

// ---- (a) VALU: lane per state, state in registers
template <int REPS>
__global__ void __launch_bounds__(64) valu_mds(const u32 *__restrict__ st_in, const u32 *__restrict__ mds, u32 *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    fr s[T];
#pragma unroll
    for (int e = 0; e < T; e++)
#pragma unroll
        for (int k = 0; k < 9; k++) s[e].l[k] = st_in[(i * T + e) * 9 + k];
    for (int rep = 0; rep < REPS; rep++) {
        fr o[T];
#pragma unroll
        for (int r = 0; r < T; r++) {
            const u32 *row = mds + (size_t)r * T * 9;
            o[r] = fr_add(fr_add(fr_dotc<6>(s, row), fr_dotc<6>(s + 6, row + 6 * 9)), fr_dotc<5>(s + 12, row + 12 * 9));
        }
#pragma unroll
        for (int r = 0; r < T; r++) s[r] = o[r];
    }
#pragma unroll
    for (int e = 0; e < T; e++)
#pragma unroll
        for (int k = 0; k < 9; k++) out[(i * T + e) * 9 + k] = s[e].l[k];
}

// ---- (b) MFMA
// canonical value (9 x 29-bit limbs) -> 32 balanced base-256 digits packed in 8 dwords
__device__ __forceinline__ void to_digits(const fr &x, u32 *d) {
    u64 w[4];
    fr_to_u64(x, w);
    u32 v[8] = {(u32)w[0], (u32)(w[0] >> 32), (u32)w[1], (u32)(w[1] >> 32), (u32)w[2], (u32)(w[2] >> 32), (u32)w[3], (u32)(w[3] >> 32)};
    u32 carry = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const u64 t = (u64)v[k] + 0x80808080u + carry;
        d[k] = (u32)t ^ 0x80808080u;
        carry = (u32)(t >> 32);
    }
}
// fold the 2 x 16 column sums of this lane into 29-bit positions (h = 0 numbering; the h = 1 lanes' limbs sit 32 bits higher)
__device__ __forceinline__ void recombine(const i32x16 &s0, const i32x16 &s1, i64 *L) {
#pragma unroll
    for (int j = 0; j < 18; j++) L[j] = 0;
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const i32 v = blk ? s1[reg] : s0[reg];
            const int k0 = 32 * blk + (reg & 3) + 8 * (reg >> 2);
            const int j0 = (8 * k0) / 29, sh0 = (8 * k0) % 29;
            L[j0] += (i64)v * (i64)(1 << sh0);
        }
}
__device__ __forceinline__ void swap32(u32 &x, u32 &y) {          // x' = (x_lo, y_lo), y' = (x_hi, y_hi) over the two halves of the wave
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}
// lo[j] at positions 29 j, hi[j] at positions 29 j + 32 = 29 (j + 1) + 3: signed column sums of a value < 17 r^2 -> value / R mod r, canonical
__device__ __forceinline__ fr reduce_columns(const i64 *lo, const i64 *hi) {
    u32 m[9], t[9];
    i64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        acc += lo[k];
        if (k >= 1) acc += hi[k - 1] << 3;
#pragma unroll
        for (int i = 0; i < k; i++) acc += (i64)((u64)m[i] * fr_p(k - i));
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (i64)((u64)m[k] * fr_p(0));
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 18; k++) {
        acc += lo[k] + (hi[k - 1] << 3);
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (i64)((u64)m[i] * fr_p(k - i));
        if (k < 17) {
            t[k - 9] = (u32)acc & FR_MASK;
            acc >>= FR_B;
        }
    }
    acc += hi[17] << 3 << FR_B;           // position 29 * 18 + 3: zero for a value in range, kept for exactness of the check
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}

template <int REPS>
__global__ void __launch_bounds__(64) mfma_mds(const u32 *__restrict__ st_in, const i32x4 *__restrict__ a_frag, u32 *__restrict__ out, size_t n_groups) {
    const int lane = threadIdx.x, h = lane >> 5, col = lane & 31;
    const size_t grp = blockIdx.x;
    if (grp >= n_groups) return;
    const size_t st = grp * 32 + col;
    // own elements: c with (c & 1) == h  (h = 0: 9 of them, h = 1: 8; slot q <-> element 2 q + h)
    fr own[9];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int c = 2 * q + h;
#pragma unroll
        for (int k = 0; k < 9; k++) own[q].l[k] = c < T ? st_in[(st * T + c) * 9 + k] : 0u;
    }
    for (int rep = 0; rep < REPS; rep++) {
        // digits of the own elements -> B fragments of all 17 (four swaps per element pair)
        i32x4 B[T + 1];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            u32 d[8];
            to_digits(own[q], d);
#pragma unroll
            for (int i = 0; i < 4; i++) swap32(d[i], d[4 + i]);
            B[2 * q] = i32x4{(i32)d[0], (i32)d[1], (i32)d[2], (i32)d[3]};
            B[2 * q + 1] = i32x4{(i32)d[4], (i32)d[5], (i32)d[6], (i32)d[7]};
        }
        // outputs in pairs (r even: owned by half 0, r + 1: by half 1)
#pragma unroll
        for (int q = 0; q < 9; q++) {
            i64 L0[18], L1[18];
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const int r = 2 * q + which;
                i32x16 s0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0;
                if (r < T) {
                    const i32x4 *ar = a_frag + ((size_t)r * T * 2) * 64 + lane;
#pragma unroll
                    for (int c = 0; c < T; c++) {
                        s0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[(c * 2) * 64], B[c], s0, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ar[(c * 2 + 1) * 64], B[c], s1, 0, 0, 0);
                    }
                }
                recombine(s0, s1, which ? L1 : L0);
            }
            // after the swaps every lane holds, for the output IT owns, the half-0 sums in L0 and the half-1 sums in L1
#pragma unroll
            for (int j = 0; j < 18; j++) {
                u32 a_lo = (u32)L0[j], a_hi = (u32)((u64)L0[j] >> 32), b_lo = (u32)L1[j], b_hi = (u32)((u64)L1[j] >> 32);
                swap32(a_lo, b_lo);
                swap32(a_hi, b_hi);
                L0[j] = (i64)(((u64)a_hi << 32) | a_lo);
                L1[j] = (i64)(((u64)b_hi << 32) | b_lo);
            }
            own[q] = reduce_columns(L0, L1);
        }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int c = 2 * q + h;
        if (c < T) {
#pragma unroll
            for (int k = 0; k < 9; k++) out[(st * T + c) * 9 + k] = own[q].l[k];
        }
    }
}

// (c) as (b), but the A fragments are BUILT, not loaded: fragment byte j of lane (row, h), block blk is digit[32 blk + row - 16 h - j] -- sixteen
// consecutive bytes of the constant's REVERSED digit string (zero-padded to 96 bytes) at a lane-dependent byte offset.  The 289 strings (27 KB)
// sit in LDS; a fragment = five aligned dword reads + four v_alignbyte_b32.  No per-layer traffic (the full fragments are 578 KB per matrix).
__device__ __forceinline__ i32x4 build_frag(const u32 *rec /* LDS, 24 dwords */, int off) {
    const u32 *p = rec + (off >> 2);
    const u32 w0 = p[0], w1 = p[1], w2 = p[2], w3 = p[3], w4 = p[4];
    const u32 sh = (u32)off & 3u;
    return i32x4{(i32)__builtin_amdgcn_alignbyte(w1, w0, sh), (i32)__builtin_amdgcn_alignbyte(w2, w1, sh), (i32)__builtin_amdgcn_alignbyte(w3, w2, sh),
                 (i32)__builtin_amdgcn_alignbyte(w4, w3, sh)};
}
template <int REPS>
__global__ void __launch_bounds__(256) mfma_mds_lds(const u32 *__restrict__ st_in, const u32 *__restrict__ rev_tab, u32 *__restrict__ out, size_t n_groups) {
    __shared__ u32 tab[T * T * 24 + 8];
    for (int i = threadIdx.x; i < T * T * 24; i += 256) tab[i] = rev_tab[i];
    if (threadIdx.x < 8) tab[T * T * 24 + threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5, col = lane & 31;
    const size_t grp = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (grp >= n_groups) return;
    const size_t st = grp * 32 + col;
    const int off0 = 63 - (col - 16 * h), off1 = off0 - 32;      // byte offsets of the two row blocks (row = col: the lane's row index is l & 31)
    fr own[9];
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int c = 2 * q + h;
#pragma unroll
        for (int k = 0; k < 9; k++) own[q].l[k] = c < T ? st_in[(st * T + c) * 9 + k] : 0u;
    }
    for (int rep = 0; rep < REPS; rep++) {
        i32x4 B[T + 1];
#pragma unroll
        for (int q = 0; q < 9; q++) {
            u32 d[8];
            to_digits(own[q], d);
#pragma unroll
            for (int i = 0; i < 4; i++) swap32(d[i], d[4 + i]);
            B[2 * q] = i32x4{(i32)d[0], (i32)d[1], (i32)d[2], (i32)d[3]};
            B[2 * q + 1] = i32x4{(i32)d[4], (i32)d[5], (i32)d[6], (i32)d[7]};
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            i64 L0[18], L1[18];
#pragma unroll
            for (int which = 0; which < 2; which++) {
                const int r = 2 * q + which;
                i32x16 s0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0;
                if (r < T) {
#pragma unroll
                    for (int c = 0; c < T; c++) {
                        const u32 *rec = tab + (r * T + c) * 24;
                        s0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(build_frag(rec, off0), B[c], s0, 0, 0, 0);
                        s1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(build_frag(rec, off1), B[c], s1, 0, 0, 0);
                    }
                }
                recombine(s0, s1, which ? L1 : L0);
            }
#pragma unroll
            for (int j = 0; j < 18; j++) {
                u32 a_lo = (u32)L0[j], a_hi = (u32)((u64)L0[j] >> 32), b_lo = (u32)L1[j], b_hi = (u32)((u64)L1[j] >> 32);
                swap32(a_lo, b_lo);
                swap32(a_hi, b_hi);
                L0[j] = (i64)(((u64)a_hi << 32) | a_lo);
                L1[j] = (i64)(((u64)b_hi << 32) | b_lo);
            }
            own[q] = reduce_columns(L0, L1);
        }
    }
#pragma unroll
    for (int q = 0; q < 9; q++) {
        const int c = 2 * q + h;
        if (c < T) {
#pragma unroll
            for (int k = 0; k < 9; k++) out[(st * T + c) * 9 + k] = own[q].l[k];
        }
    }
}

// ---- host
static u64 rng_state = 0x9E3779B97F4A7C15ULL;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static fr rand_fr() {
    u64 w[4] = {rnd(), rnd(), rnd(), rnd() & 0x0FFFFFFFFFFFFFFFULL};   // < 2^252 < r
    return fr_from_u64(w);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = (size_t)1 << 19;
    constexpr int REPS = 8;
    std::vector<u32> mds(T * T * 9), st(n * T * 9);
    std::vector<fr> M(T * T);
    for (int i = 0; i < T * T; i++) { M[i] = rand_fr(); memcpy(&mds[i * 9], M[i].l, 36); }
    for (size_t i = 0; i < n * T; i++) { const fr x = rand_fr(); memcpy(&st[i * 9], x.l, 36); }
    // A fragments: constant (r, c), block blk, lane l (row = l & 31, half h): bytes A[k = 32 blk + row][i = 16 h + j] = digit[k - i]
    std::vector<signed char> af((size_t)T * T * 2 * 64 * 16);
    for (int rc = 0; rc < T * T; rc++) {
        u64 w[4];
        fr_to_u64(M[rc], w);
        int dg[33], carry = 0;
        for (int i = 0; i < 32; i++) { int d = (int)((w[i / 8] >> (8 * (i % 8))) & 0xFF) + carry; carry = 0; if (d >= 128) { d -= 256; carry = 1; } dg[i] = d; }
        if (carry) { printf("constant needs a 33rd digit\n"); return 1; }
        for (int blk = 0; blk < 2; blk++) for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
            const int k = 32 * blk + (l & 31), i = 16 * (l >> 5) + j, d = k - i;
            af[(((size_t)rc * 2 + blk) * 64 + l) * 16 + j] = (signed char)((d >= 0 && d < 32) ? dg[d] : 0);
        }
    }
    // reversed, zero-padded digit strings: bytes [32, 64) = digit[31 - u]; a fragment starts at byte 63 - (32 blk + row - 16 h)
    std::vector<signed char> rev((size_t)T * T * 96, 0);
    for (int rc = 0; rc < T * T; rc++) {
        u64 w[4];
        fr_to_u64(M[rc], w);
        int carry = 0;
        for (int i = 0; i < 32; i++) { int d = (int)((w[i / 8] >> (8 * (i % 8))) & 0xFF) + carry; carry = 0; if (d >= 128) { d -= 256; carry = 1; } rev[(size_t)rc * 96 + 32 + (31 - i)] = (signed char)d; }
    }
    u32 *d_rev, *d_o3;
    CHK(hipMalloc(&d_rev, rev.size())); CHK(hipMemcpy(d_rev, rev.data(), rev.size(), hipMemcpyHostToDevice));
    u32 *d_mds, *d_st, *d_o1, *d_o2; i32x4 *d_af;
    CHK(hipMalloc(&d_mds, mds.size() * 4)); CHK(hipMalloc(&d_st, st.size() * 4)); CHK(hipMalloc(&d_o1, st.size() * 4)); CHK(hipMalloc(&d_o2, st.size() * 4));
    CHK(hipMalloc(&d_af, af.size())); CHK(hipMalloc(&d_o3, st.size() * 4));
    CHK(hipMemcpy(d_mds, mds.data(), mds.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_st, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_af, af.data(), af.size(), hipMemcpyHostToDevice));
    // correctness: one application both ways, and against the host's fr_mul / fr_add for a few states
    hipLaunchKernelGGL(valu_mds<1>, dim3(n / 64), dim3(64), 0, 0, d_st, d_mds, d_o1, n);
    hipLaunchKernelGGL(mfma_mds<1>, dim3(n / 32), dim3(64), 0, 0, d_st, d_af, d_o2, n / 32);
    CHK(hipDeviceSynchronize());
    std::vector<u32> o1(st.size()), o2(st.size());
    CHK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, bad_host = 0;
    for (size_t i = 0; i < o1.size(); i++) bad += o1[i] != o2[i];
    for (size_t s = 0; s < n; s += 4099)
        for (int r = 0; r < T; r++) {
            fr acc = fr_zero();
            for (int c = 0; c < T; c++) { fr x; memcpy(x.l, &st[(s * T + c) * 9], 36); acc = fr_add(acc, fr_mul(M[r * T + c], x)); }
            for (int k = 0; k < 9; k++) bad_host += acc.l[k] != o2[(s * T + r) * 9 + k];
        }
    printf("%zu states x 17 outputs: MFMA vs VALU mismatching words %zu, MFMA vs host definition %zu\n", n, bad, bad_host);
    hipLaunchKernelGGL(mfma_mds_lds<1>, dim3(n / 128), dim3(256), 0, 0, d_st, d_rev, d_o3, n / 32);
    CHK(hipDeviceSynchronize());
    std::vector<u32> o3(st.size());
    CHK(hipMemcpy(o3.data(), d_o3, o3.size() * 4, hipMemcpyDeviceToHost));
    size_t bad3 = 0;
    for (size_t i = 0; i < o1.size(); i++) bad3 += o1[i] != o3[i];
    printf("fragments built from the LDS table: mismatching words %zu\n", bad3);
    bad += bad3;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(valu_mds<REPS>, dim3(n / 64), dim3(64), 0, 0, d_st, d_mds, d_o1, n);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) VALU  17 x 17 constant matrix x state: %7.3f ms  %7.2f M layers/s\n", ms, (double)n * REPS / ms / 1e3);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_mds<REPS>, dim3(n / 32), dim3(64), 0, 0, d_st, d_af, d_o2, n / 32);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) MFMA  17 x 17 constant matrix x state: %7.3f ms  %7.2f M layers/s\n", ms, (double)n * REPS / ms / 1e3);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_mds_lds<REPS>, dim3(n / 128), dim3(256), 0, 0, d_st, d_rev, d_o3, n / 32);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(c) MFMA, fragments from an LDS table     : %7.3f ms  %7.2f M layers/s\n", ms, (double)n * REPS / ms / 1e3);
    }
    CHK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost)); CHK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    size_t bad8 = 0;
    for (size_t i = 0; i < o1.size(); i++) bad8 += o1[i] != o2[i];
    printf("after %d chained layers: mismatching words %zu\n", REPS, bad8);
    return (bad || bad_host || bad8) ? 1 : 0;
}
