// The 12 x 12 small-constant linear layer of Poseidon-Goldilocks on the matrix cores, gfx950 (measurement tool).
//
// out_r = sum_c M[r][c] s_c + k_r over 64-bit values with M[r][c] <= 49: the production form (csrc/poseidon_mds_asm.inc) is 24 v_mad_u64_u32
// per row over the 32-bit halves of the state + a 4-instruction weak fold -- 59 % of a permutation's 17.5 k instructions.  Here the byte
// products go to v_mfma_i32_4x4x4i8 (sixteen independent 4 x 4 x 4 blocks per instruction): lane = state, the B operand of a lane is ONE dword
// holding byte d of four consecutive state elements (a 4 x 4 byte transpose of the state's dwords: 8 v_perm_b32 per four elements and half,
// bytes biased by -128 for the signed operand, the bias a constant folded into k_r), the A operand a 4 x 4 block of M; 3 blocks of rows x 3 of
// columns x 8 byte positions = 72 instructions per layer give, per row and half, four byte-position sums that three v_mad_i64_i32 put back
// into the SAME two 64-bit sums A, B the production form folds.  Per row: 8 multiply-adds + the fold instead of 24 + the fold, plus the
// transposes (6 per row).  Same words as the production layer (both canonicalised), checked over all states.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I eigen_zeth_amd/csrc -o tools/ubench_mfma_gl_mds tools/ubench_mfma_gl_mds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define GL_ASM_SCRATCH_BASE 52
#include "gl.hpp"
#include "gl_asm.hpp"
typedef long long i64;
typedef int i32;
typedef i32 v4i __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ constexpr u32 def_mds(int i, int j) {
    constexpr u32 circ[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
    return circ[(j - i + 12) % 12] + ((i == 0 && j == 0) ? 8u : 0u);
}
#include "poseidon_mds_asm.inc"

template <int REPS>
__global__ void __launch_bounds__(256) valu_layer(u64 *st, const u64 *__restrict__ c, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = st[j * n + i];
    for (int rep = 0; rep < REPS; rep++) mds_ark_default_asm<true>(s, c);
#pragma unroll
    for (int j = 0; j < 12; j++) st[j * n + i] = gl_canon(s[j]);
}

// A + B 2^32 (both < 2^44) -> weak u64: the closing four instructions of the production rows (poseidon_mds_asm.inc)
__device__ __forceinline__ u64 fold_weak(u64 A, u64 B) {
    const u32 b0 = (u32)B, b1 = (u32)(B >> 32);
    u32 r0, r1;
    u64 sd, c, d;
    asm("v_mad_u64_u32 " GL_P0 ", %2, %6, -1, %5\n\t"          // T = A + b1 * EPS
        "v_add_co_u32 %1, %3, " GL_V1 ", %7\n\t"               // hi = T_hi + b0           -> c
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %0, %4, 0, " GL_V0 ", %3\n\t"        // lo = T_lo - c            -> d
        "s_nop 1\n\t"
        "s_andn2_b64 %3, %3, %4\n\t"
        "v_addc_co_u32 %1, %4, %1, 0, %3"                      // hi += c & ~d
        : "=&v"(r0), "=&v"(r1), "=&s"(sd), "=&s"(c), "=&s"(d)
        : "v"(A), "v"(b1), "v"(b0)
        : "scc", "" GL_V0 "", "" GL_V1 "");
    return ((u64)r1 << 32) | r0;
}
// acc = x * m + acc with m = 2^(8 d) held in an SGPR the optimiser cannot see through: it then keeps the signed 32 x 32 + 64 multiply-add
// (for a literal power of two it emits a sign extension and a 64-bit shift-add instead).  Not inline asm: the hazard recogniser does not pad
// between a matrix instruction and an inline-asm reader of its result registers (measured: wrong words in the rows scheduled next to them).
__device__ __forceinline__ i32 opaque(i32 v) {
    asm("" : "+s"(v));
    return v;
}
template <int REPS>
__global__ void __launch_bounds__(256) mfma_layer(u64 *st, const u64 *__restrict__ c, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = i < n;
    const size_t ii = on ? i : 0;
    u64 s[12];
#pragma unroll
    for (int j = 0; j < 12; j++) s[j] = st[j * n + ii];
    // this lane's row of every 4 x 4 block of M (row = lane & 3)
    const int row = threadIdx.x & 3;
    i32 ablk[3][3];
#pragma unroll
    for (int R = 0; R < 3; R++)
#pragma unroll
        for (int C = 0; C < 3; C++) {
            u32 w = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                u32 m = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) m = row == r ? def_mds(4 * R + r, 4 * C + k) : m;
                w |= m << (8 * k);
            }
            ablk[R][C] = (i32)w;
        }
    // k_r = next round constant + the bias of the signed bytes: 128 * rowsum_r * 0x01010101 per half
    u64 klo[12], khi[12];
#pragma unroll
    for (int r = 0; r < 12; r++) {
        u32 rs = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) rs += def_mds(r, j);
        const u64 bias = (u64)(128u * rs) * 0x01010101ULL;
        klo[r] = (u64)(u32)c[r] + bias;
        khi[r] = (c[r] >> 32) + bias;
    }
    const i32 mul[4] = {opaque(1), opaque(1 << 8), opaque(1 << 16), opaque(1 << 24)};
    for (int rep = 0; rep < REPS; rep++) {
        i32 bd[3][2][4];
#pragma unroll
        for (int C = 0; C < 3; C++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const u32 w0 = (u32)(s[4 * C] >> (32 * h)), w1 = (u32)(s[4 * C + 1] >> (32 * h)), w2 = (u32)(s[4 * C + 2] >> (32 * h)), w3 = (u32)(s[4 * C + 3] >> (32 * h));
                const u32 a = __builtin_amdgcn_perm(w1, w0, 0x05010400u), b = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
                const u32 cc = __builtin_amdgcn_perm(w3, w2, 0x05010400u), d = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
                bd[C][h][0] = (i32)(__builtin_amdgcn_perm(cc, a, 0x05040100u) ^ 0x80808080u);
                bd[C][h][1] = (i32)(__builtin_amdgcn_perm(cc, a, 0x07060302u) ^ 0x80808080u);
                bd[C][h][2] = (i32)(__builtin_amdgcn_perm(d, b, 0x05040100u) ^ 0x80808080u);
                bd[C][h][3] = (i32)(__builtin_amdgcn_perm(d, b, 0x07060302u) ^ 0x80808080u);
            }
        u64 o[12];
#pragma unroll
        for (int R = 0; R < 3; R++) {
            v4i acc[2][4];
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    v4i z = {0, 0, 0, 0};
#pragma unroll
                    for (int C = 0; C < 3; C++) z = __builtin_amdgcn_mfma_i32_4x4x4i8(ablk[R][C], bd[C][h][d], z, 0, 0, 0);
                    acc[h][d] = z;
                }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                i64 A = (i64)klo[4 * R + r], B = (i64)khi[4 * R + r];
#pragma unroll
                for (int d = 0; d < 4; d++) {
                    A += (i64)acc[0][d][r] * (i64)mul[d];
                    B += (i64)acc[1][d][r] * (i64)mul[d];
                }
                o[4 * R + r] = fold_weak((u64)A, (u64)B);
            }
        }
#pragma unroll
        for (int j = 0; j < 12; j++) s[j] = o[j];
    }
    if (on) {
#pragma unroll
        for (int j = 0; j < 12; j++) st[j * n + i] = gl_canon(s[j]);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const size_t n = (size_t)1 << 22;
    constexpr int REPS = 30;
    std::vector<u64> h(12 * n), cst(12);
    u64 x = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (auto &v : h) v = rnd();                 // any u64: the layer takes weak values
    for (int e = 0; e < 16; e++) h[e] = e == 0 ? 0 : e == 1 ? ~0ULL : e == 2 ? GL_P : e == 3 ? GL_P - 1 : (0x8080808080808080ULL >> e);
    for (auto &v : cst) v = rnd() % GL_P;
    u64 *d_a, *d_b, *d_c;
    CHK(hipMalloc(&d_a, h.size() * 8)); CHK(hipMalloc(&d_b, h.size() * 8)); CHK(hipMalloc(&d_c, 96));
    CHK(hipMemcpy(d_c, cst.data(), 96, hipMemcpyHostToDevice));
    auto reset = [&]() { CHK(hipMemcpy(d_a, h.data(), h.size() * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(d_b, h.data(), h.size() * 8, hipMemcpyHostToDevice)); };
    reset();
    hipLaunchKernelGGL(valu_layer<1>, dim3(n / 256), dim3(256), 0, 0, d_a, d_c, n);
    hipLaunchKernelGGL(mfma_layer<1>, dim3(n / 256), dim3(256), 0, 0, d_b, d_c, n);
    CHK(hipDeviceSynchronize());
    std::vector<u64> ra(h.size()), rb(h.size());
    CHK(hipMemcpy(ra.data(), d_a, ra.size() * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(rb.data(), d_b, rb.size() * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, bad_def = 0;
    for (size_t k = 0; k < ra.size(); k++) bad += ra[k] != rb[k];
    for (size_t s = 0; s < n; s += 65537)        // against the definition on the host
        for (int r = 0; r < 12; r++) {
            unsigned __int128 acc = cst[r];
            for (int j = 0; j < 12; j++) acc += (unsigned __int128)def_mds(r, j) * h[j * n + s];
            bad_def += (u64)(acc % GL_P) != rb[r * n + s];
        }
    printf("%zu states x 12 outputs, one layer: MFMA vs production words differing %zu, MFMA vs definition %zu\n", n, bad, bad_def);
    if (bad) {
        size_t byrow[12] = {0};
        for (int r = 0; r < 12; r++) for (size_t s = 0; s < n; s++) byrow[r] += ra[r * n + s] != rb[r * n + s];
        printf("   by row:"); for (int r = 0; r < 12; r++) printf(" %zu", byrow[r]); printf("\n");
        for (int r = 0; r < 12; r++) if (byrow[r]) { for (size_t s = 0; s < 3; s++) printf("   row %d state %zu: production %016llx mfma %016llx\n", r, s, ra[r * n + s], rb[r * n + s]); break; }
    }
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        reset();
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(valu_layer<REPS>, dim3(n / 256), dim3(256), 0, 0, d_a, d_c, n);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(a) production (24 mads per row, asm) : %7.3f ms  %7.2f G layers/s\n", ms, (double)n * REPS / ms / 1e6);
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_layer<REPS>, dim3(n / 256), dim3(256), 0, 0, d_b, d_c, n);
        CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize()); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("(b) MFMA 4x4x4 i8 + 8 mads per row    : %7.3f ms  %7.2f G layers/s\n", ms, (double)n * REPS / ms / 1e6);
    }
    CHK(hipMemcpy(ra.data(), d_a, ra.size() * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(rb.data(), d_b, rb.size() * 8, hipMemcpyDeviceToHost));
    size_t bad30 = 0;
    for (size_t k = 0; k < ra.size(); k++) bad30 += ra[k] != rb[k];
    printf("after %d chained layers: words differing %zu\n", REPS, bad30);
    return (bad || bad_def || bad30) ? 1 : 0;
}
