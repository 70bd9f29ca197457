// MFMA vs VALU for the limb products of a 256-bit field multiplication (BN254 F_q / F_r), gfx950 (measurement tool).
//
// The question (north_star: "MFMA where a limb-product formulation maps to matrix cores"): can the i8 matrix cores take
// the schoolbook limb products of msm.hip's field multiplication?  An MFMA multiplies ONE A matrix with 32 B columns, so
// it only helps where one operand is shared by many products: the two constant multiplications of a Montgomery reduction
// (by q' and by q).  With per-lane operand pairs (bucket accumulation) every product would need its own Toeplitz A matrix,
// i.e. 1 useful column of 32 -- not measured, it is 32x worse than what is measured here.
//
// Measured: t = m * q for many m and ONE constant q.
//   (a) VALU: 9 x 9 limbs of 29 bits, 81 v_mad_u64_u32 per product per lane into 17 unnormalised 64-bit columns
//       (what msm.hip's fq_mul spends on this half of a Montgomery multiplication).
//   (b) MFMA: m as 32 balanced base-256 digits in [-128,127] (prepared on the host: the conversion is NOT timed, in the
//       MFMA's favour), A = Toeplitz(q digits) as two 32x32 blocks, two v_mfma_i32_32x32x32_i8 per 32 products, then the
//       64 signed column sums (|c_k| < 2^19) are folded into 29-bit-position 64-bit limbs with one v_mad_i64_i32 each.
//       The two half-lanes that share a column write their partial limbs separately (the cross-lane add is NOT timed either).
// Both results are checked on the host: sum_j limb_j 2^(29 j) == m * q exactly.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_mfma_fq tools/ubench_mfma_fq.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;
typedef int i32;
typedef __int128 i128;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef i32 i32x16 __attribute__((ext_vector_type(16)));
typedef i32 i32x4 __attribute__((ext_vector_type(4)));

struct QLimbs { u32 q[9]; };

// (a) one product per lane: columns c[k] = sum_{i+j=k} m_i q_j, 81 mads.  REPS products per lane, chained through the input.
template <int REPS>
__global__ void __launch_bounds__(256) valu_mulq(const u32 *__restrict__ m_in, u64 *__restrict__ cols_out, QLimbs Q, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u32 m[9];
#pragma unroll
    for (int j = 0; j < 9; j++) m[j] = m_in[i * 9 + j];
    u64 c[17];
    for (int r = 0; r < REPS; r++) {
#pragma unroll
        for (int k = 0; k < 17; k++) c[k] = 0;
#pragma unroll
        for (int a = 0; a < 9; a++)
#pragma unroll
            for (int b = 0; b < 9; b++) c[a + b] += (u64)m[a] * Q.q[b];
        if (r + 1 < REPS) {
#pragma unroll
            for (int j = 0; j < 9; j++) m[j] = (m[j] ^ (u32)c[j + 4]) & 0x1FFFFFFFu;   // next operand depends on this product
        }
    }
#pragma unroll
    for (int k = 0; k < 17; k++) cols_out[i * 17 + k] = c[k];
}

// (b) one wave = 32 products per step.  a_frag: [2 blocks][64 lanes] x 16 bytes (Toeplitz of q's digits), md: n x 32 digits.
// lane l: column c = l & 31, half h = l >> 5; B fragment = digits 16h .. 16h+15 of value c.
// C/D map (32x32): row = (reg & 3) + 8 (reg >> 2) + 4 h, col = l & 31.
template <int REPS>
__global__ void __launch_bounds__(256) mfma_mulq(const i32x4 *__restrict__ a_frag, const i32x4 *__restrict__ md, i64 *__restrict__ limbs_out, size_t n_groups) {
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    const size_t grp = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6;      // one group of 32 values per wave
    if (grp >= n_groups) return;
    const i32x4 a0 = a_frag[lane], a1 = a_frag[64 + lane];
    i32x4 b = md[(grp * 32 + c) * 2 + h];
    i64 L[18];
    for (int r = 0; r < REPS; r++) {
        i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const i32x16 s0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, b, z, 0, 0, 0);
        const i32x16 s1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, b, z, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < 18; j++) L[j] = 0;
        // rows of this lane: k = 32 blk + (reg & 3) + 8 (reg >> 2) + 4 h.  The h = 1 lanes sit 32 bits higher: their limb
        // array is read as positions 29 j + 32, so both halves run the same code: row k0 (h = 0 numbering) -> limb (8 k0) / 29,
        // shift (8 k0) % 29, ONE v_mad_i64_i32 per column sum (multiplier 2^shift in an SGPR).
#pragma unroll
        for (int blk = 0; blk < 2; blk++)
#pragma unroll
            for (int reg = 0; reg < 16; reg++) {
                const i32 v = blk ? s1[reg] : s0[reg];
                const int k0 = 32 * blk + (reg & 3) + 8 * (reg >> 2);
                const int j0 = (8 * k0) / 29, sh0 = (8 * k0) % 29;
                u64 sd;
                asm("v_mad_i64_i32 %0, %1, %2, %3, %0" : "+v"(L[j0]), "=s"(sd) : "v"(v), "s"(1 << sh0));
            }
        if (r + 1 < REPS) {                                                 // next operand depends on this product
            b[0] = (b[0] ^ (i32)L[3]) & 0x3F3F3F3F; b[1] = (b[1] ^ (i32)L[5]) & 0x3F3F3F3F;
            b[2] = (b[2] ^ (i32)L[7]) & 0x3F3F3F3F; b[3] = (b[3] ^ (i32)L[9]) & 0x3F3F3F3F;
        }
    }
#pragma unroll
    for (int j = 0; j < 18; j++) limbs_out[((grp * 32 + c) * 2 + h) * 18 + j] = L[j];
}

static u64 rng_state = 0x9E3779B97F4A7C15ULL;
static u64 rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

// 512-bit helpers on the host (little-endian 32-bit words)
struct Big { u32 w[20]; };
static Big big_mul(const u32 *a, const u32 *b) {   // 8 x 8 words
    Big r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 8; i++) {
        u64 carry = 0;
        for (int j = 0; j < 8; j++) { u64 t = (u64)a[i] * b[j] + r.w[i + j] + carry; r.w[i + j] = (u32)t; carry = t >> 32; }
        r.w[i + 8] = (u32)carry;
    }
    return r;
}
// sum_j limb[j] * 2^(pos_bits * j) with signed 128-bit-safe accumulation, compared to big
static bool check_positional(const i64 *limb, int nl, int pos_bits, const Big &want) {
    // accumulate into a signed array of 32-bit words with carries
    i128 acc[24]; for (auto &x : acc) x = 0;
    for (int j = 0; j < nl; j++) {
        const int bit = pos_bits * j, w = bit / 32, sh = bit % 32;
        acc[w] += ((i128)limb[j]) << sh;
    }
    i128 carry = 0;
    for (int w = 0; w < 20; w++) {
        i128 t = acc[w] + carry;
        u32 lo = (u32)(t & 0xFFFFFFFF);
        carry = (t - lo) >> 32;
        if (lo != want.w[w]) return false;
    }
    return carry == 0;
}

// MFMA result: half 0 limbs at positions 29 j, half 1 limbs at positions 29 j + 32
static bool check_two_halves(const i64 *l0, const i64 *l1, const Big &want) {
    i128 acc[24]; for (auto &x : acc) x = 0;
    for (int j = 0; j < 18; j++) {
        int bit = 29 * j; acc[bit / 32] += ((i128)l0[j]) << (bit % 32);
        bit += 32;        acc[bit / 32] += ((i128)l1[j]) << (bit % 32);
    }
    i128 carry = 0;
    for (int w = 0; w < 20; w++) {
        i128 t = acc[w] + carry;
        u32 lo = (u32)(t & 0xFFFFFFFF);
        carry = (t - lo) >> 32;
        if (lo != want.w[w]) return false;
    }
    return carry == 0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = 1 << 20;                       // products per launch (x REPS)
    constexpr int REPS = 16;
    // constant q: the BN254 base-field modulus
    const u32 qw[8] = {0xd87cfd47, 0x3c208c16, 0x6871ca8d, 0x97816a91, 0x8181585d, 0xb85045b6, 0xe131a029, 0x30644e72};
    QLimbs Q;
    for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, w = bit / 32, sh = bit % 32;
        u64 v = (u64)qw[w] >> sh; if (w + 1 < 8 && sh) v |= (u64)qw[w + 1] << (32 - sh);
        Q.q[j] = (u32)(v & 0x1FFFFFFF);
    }
    // balanced base-256 digits of q
    int qd[33]; { int carry = 0; for (int i = 0; i < 32; i++) { int d = ((qw[i / 4] >> (8 * (i % 4))) & 0xFF) + carry; carry = 0; if (d >= 128) { d -= 256; carry = 1; } qd[i] = d; } qd[32] = carry; }
    if (qd[32]) { printf("q needs a 33rd digit\n"); return 1; }
    // inputs m < 2^253 (so that balanced digits fit 32 digits): words, 29-bit limbs, balanced digits
    std::vector<u32> mw(n * 8), ml(n * 9); std::vector<signed char> mdg(n * 32);
    for (size_t i = 0; i < n; i++) {
        for (int j = 0; j < 8; j++) mw[i * 8 + j] = (u32)rnd();
        mw[i * 8 + 7] &= 0x0FFFFFFF;
        for (int j = 0; j < 9; j++) {
            const int bit = 29 * j, w = bit / 32, sh = bit % 32;
            u64 v = (u64)mw[i * 8 + w] >> sh; if (w + 1 < 8 && sh) v |= (u64)mw[i * 8 + w + 1] << (32 - sh);
            ml[i * 9 + j] = (u32)(v & 0x1FFFFFFF);
        }
        int carry = 0;
        for (int d = 0; d < 32; d++) { int v = ((mw[i * 8 + d / 4] >> (8 * (d % 4))) & 0xFF) + carry; carry = 0; if (v >= 128) { v -= 256; carry = 1; } mdg[i * 32 + d] = (signed char)v; }
        if (carry) { printf("digit overflow\n"); return 1; }
    }
    // A fragments: block blk, lane l (row r = l & 31, half h): bytes A[k = 32 blk + r][i = 16 h + j] = qd[k - i]
    std::vector<signed char> af(2 * 64 * 16);
    for (int blk = 0; blk < 2; blk++) for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
        const int k = 32 * blk + (l & 31), i = 16 * (l >> 5) + j, d = k - i;
        af[(blk * 64 + l) * 16 + j] = (signed char)((d >= 0 && d < 32) ? qd[d] : 0);
    }
    u32 *d_ml; u64 *d_cols; i32x4 *d_af, *d_md; i64 *d_limbs;
    CHK(hipMalloc(&d_ml, n * 9 * 4)); CHK(hipMalloc(&d_cols, n * 17 * 8)); CHK(hipMalloc(&d_af, af.size())); CHK(hipMalloc(&d_md, n * 32));
    CHK(hipMalloc(&d_limbs, n * 2 * 18 * 8));
    CHK(hipMemcpy(d_ml, ml.data(), n * 9 * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_af, af.data(), af.size(), hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_md, mdg.data(), n * 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    // ---- correctness (REPS = 1)
    hipLaunchKernelGGL(valu_mulq<1>, dim3(n / 256), dim3(256), 0, 0, d_ml, d_cols, Q, n);
    hipLaunchKernelGGL(mfma_mulq<1>, dim3(n / 32 * 64 / 256), dim3(256), 0, 0, d_af, d_md, d_limbs, n / 32);
    CHK(hipDeviceSynchronize());
    std::vector<u64> cols(n * 17); std::vector<i64> limbs(n * 36);
    CHK(hipMemcpy(cols.data(), d_cols, n * 17 * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(limbs.data(), d_limbs, n * 36 * 8, hipMemcpyDeviceToHost));
    size_t bad_a = 0, bad_b = 0;
    for (size_t i = 0; i < n; i += 37) {
        const Big want = big_mul(&mw[i * 8], qw);
        i64 ca[17]; for (int k = 0; k < 17; k++) ca[k] = (i64)cols[i * 17 + k];
        if (!check_positional(ca, 17, 29, want)) bad_a++;
        if (!check_two_halves(&limbs[(i * 2) * 18], &limbs[(i * 2 + 1) * 18], want)) bad_b++;
    }
    printf("checked %zu products: VALU mismatches %zu, MFMA mismatches %zu\n", (n + 36) / 37, bad_a, bad_b);
    // ---- timing
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(valu_mulq<REPS>, dim3(n / 256), dim3(256), 0, 0, d_ml, d_cols, Q, n);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) VALU 81-mad constant product : %7.3f ms  %7.1f G products/s\n", ms, (double)n * REPS / ms / 1e6);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_mulq<REPS>, dim3(n / 32 * 64 / 256), dim3(256), 0, 0, d_af, d_md, d_limbs, n / 32);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) MFMA i8 32x32x32 + recombine : %7.3f ms  %7.1f G products/s\n", ms, (double)n * REPS / ms / 1e6);
    }
    return (bad_a || bad_b) ? 1 : 0;
}
