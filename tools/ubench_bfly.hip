// Throughput + equality check of Goldilocks butterfly formulations on gfx950 (measurement tool, not product).
//   variant 0: compiler-scheduled gl_add / gl_sub (csrc/gl.hpp)
//   variant 1: hand-scheduled carry-chain butterflies (csrc/gl_asm.hpp): 10 VALU + 2 SALU per butterfly
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_bfly tools/ubench_bfly.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../eigen_zeth_amd/csrc/gl.hpp"
#include "../eigen_zeth_amd/csrc/gl_asm.hpp"

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 64;

template <int VAR>
__global__ void __launch_bounds__(256) k(const u64 *in, u64 *out) {
    u64 v[16];
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = in[g * 16 + i];
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int l = 0; l < 4; l++) {
            const int half = 8 >> l;
            if (VAR == 0) {
#pragma unroll
                for (int i = 0; i < 16; i++)
                    if (!(i & half)) { u64 x = v[i], y = v[i + half]; v[i] = gl_add(x, y); v[i + half] = gl_sub(x, y); }
            } else {
                int idx[8], n = 0;
#pragma unroll
                for (int i = 0; i < 16; i++) if (!(i & half)) idx[n++] = i;
#pragma unroll
                for (int q = 0; q < 8; q += 2) gl_bfly2(v[idx[q]], v[idx[q] + half], v[idx[q + 1]], v[idx[q + 1] + half]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[g * 16 + i] = v[i];
}

template <int VAR>
__global__ void __launch_bounds__(256) km(const u64 *in, u64 *out) {
    u64 v[16];
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = in[g * 16 + i];
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const u64 b0 = v[(i + 5) & 15] ^ (u64)it, b1 = v[(i + 9) & 15] + 12345u;   // any u64 (also >= p)
            if (VAR == 0) { v[i] = gl_mul(v[i], b0); v[i + 1] = gl_mul(v[i + 1], b1); }
            else gl_mul2(v[i], b0, v[i + 1], b1);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[g * 16 + i] = v[i];
}

// weak products (gl_mul2w / gl_mul1w): one step per pair, operands also non-canonical (>= p, up to 2^64 - 1);
// canon(weak) must equal gl_mul
__global__ void __launch_bounds__(256) kw(const u64 *in, unsigned long long *bad) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    u64 v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = in[g * 16 + i];
    unsigned long long nb = 0;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        u64 a0 = v[i], a1 = v[i + 1];
        if ((g + i) % 7 == 0) { a0 = ~a0; a1 += GL_P - 1; }                 // weak inputs
        if ((g + i) % 11 == 0) { a0 = 0xFFFFFFFFFFFFFFFFULL; }
        const u64 b0 = v[(i + 5) & 15] ^ 0x8000000000000000ULL, b1 = v[(i + 9) & 15];
        const u64 r0 = gl_mul(a0, b0), r1 = gl_mul(a1, b1);
        u64 w0 = a0, w1 = a1;
        gl_mul2w(w0, b0, w1, b1);
        const u64 w2 = gl_mul1w(a1, b0), r2 = gl_mul(a1, b0);
        nb += (gl_canon(w0) != r0) + (gl_canon(w1) != r1) + (gl_canon(w2) != r2);
    }
    if (nb) atomicAdd(bad, nb);
}

template <int VAR>
__global__ void __launch_bounds__(256) ks(const u64 *in, u64 *out) {
    u64 v[16];
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = in[g * 16 + i];
    for (int it = 0; it < ITERS; it++) {
        if (VAR == 0) {
            v[1] = gl_mul_pow2<12>(v[1]); v[2] = gl_mul_pow2<24>(v[2]); v[3] = gl_mul_pow2<36>(v[3]); v[4] = gl_mul_pow2<48>(v[4]);
            v[5] = gl_mul_pow2<60>(v[5]); v[6] = gl_mul_pow2<72>(v[6]); v[7] = gl_mul_pow2<84>(v[7]); v[8] = gl_mul_pow2<48>(v[8]);
            v[9] = gl_mul_pow2<12>(v[9]); v[10] = gl_mul_pow2<24>(v[10]); v[11] = gl_mul_pow2<36>(v[11]); v[12] = gl_mul_pow2<48>(v[12]);
            v[13] = gl_mul_pow2<60>(v[13]); v[14] = gl_mul_pow2<72>(v[14]); v[15] = gl_mul_pow2<84>(v[15]); v[0] = gl_mul_pow2<48>(v[0]);
        } else {
            v[1] = gl_shl12<1>(v[1]); v[2] = gl_shl12<2>(v[2]); v[3] = gl_shl12<3>(v[3]); v[4] = gl_shl12<4>(v[4]);
            v[5] = gl_shl12<5>(v[5]); v[6] = gl_shl12<6>(v[6]); v[7] = gl_shl12<7>(v[7]); v[8] = gl_shl12<4>(v[8]);
            v[9] = gl_shl12<1>(v[9]); v[10] = gl_shl12<2>(v[10]); v[11] = gl_shl12<3>(v[11]); v[12] = gl_shl12<4>(v[12]);
            v[13] = gl_shl12<5>(v[13]); v[14] = gl_shl12<6>(v[14]); v[15] = gl_shl12<7>(v[15]); v[0] = gl_shl12<4>(v[0]);
        }
    }
#pragma unroll
    for (int i = 0; i < 16; i++) out[g * 16 + i] = v[i];
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("start\n");
    const int blocks = 256 * 8;
    const size_t n = (size_t)blocks * 256 * 16;
    std::vector<u64> h(n);
    u64 s = 88172645463325252ULL;
    for (size_t i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        u64 v = s;
        if (i % 97 == 0) v = GL_P - 1 - (i % 5);         // stress values next to p
        if (i % 101 == 0) v = (i % 3);                   // and next to 0
        if (i % 103 == 0) v = 0xFFFFFFFF00000000ULL - (i % 2);
        h[i] = v >= GL_P ? v - GL_P : v;
    }
    u64 *d_in, *d0, *d1;
    CHK(hipMalloc(&d_in, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8));
    CHK(hipMemcpy(d_in, h.data(), n * 8, hipMemcpyHostToDevice));
    printf("uploaded\n");
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms[2];
    for (int var = 0; var < 2; var++) {
        u64 *o = var ? d1 : d0;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            CHK(hipEventElapsedTime(&ms[var], e0, e1));
        }
        const double bf = (double)blocks * 256 * ITERS * 32;
        printf("variant %d: %.3f ms  %.1f G butterflies/s  (%.2f T add|sub /s)\n", var, ms[var], bf / ms[var] / 1e6, 2 * bf / ms[var] / 1e9);
    }
    for (int var = 0; var < 2; var++) {
        u64 *o = var ? d1 : d0;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(km<0>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            else hipLaunchKernelGGL(km<1>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            CHK(hipEventElapsedTime(&ms[var], e0, e1));
        }
        const double mu = (double)blocks * 256 * ITERS * 16;
        printf("mul variant %d: %.3f ms  %.2f T modmul/s\n", var, ms[var], mu / ms[var] / 1e9);
    }
    {
        std::vector<u64> m0(n), m1(n);
        CHK(hipMemcpy(m0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(m1.data(), d1, n * 8, hipMemcpyDeviceToHost));
        size_t badm = 0, nc = 0;
        for (size_t i = 0; i < n; i++) { if (m0[i] != m1[i]) badm++; if (m1[i] >= GL_P) nc++; }
        printf("gl_mul2 mismatches vs gl_mul: %zu of %zu; non-canonical: %zu\n", badm, n, nc);
        if (badm || nc) return 1;
    }
    {
        unsigned long long *d_bad, h_bad = 0;
        CHK(hipMalloc(&d_bad, 8)); CHK(hipMemset(d_bad, 0, 8));
        hipLaunchKernelGGL(kw, dim3(blocks), dim3(256), 0, 0, d_in, d_bad);
        CHK(hipMemcpy(&h_bad, d_bad, 8, hipMemcpyDeviceToHost));
        printf("weak products gl_mul2w / gl_mul1w: canon(weak) != gl_mul in %llu of %zu\n", h_bad, (size_t)blocks * 256 * 24);
        if (h_bad) return 1;
    }
    for (int var = 0; var < 2; var++) {
        u64 *o = var ? d1 : d0;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            if (var == 0) hipLaunchKernelGGL(ks<0>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            else hipLaunchKernelGGL(ks<1>, dim3(blocks), dim3(256), 0, 0, d_in, o);
            CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
            CHK(hipEventElapsedTime(&ms[var], e0, e1));
        }
        const double mu = (double)blocks * 256 * ITERS * 16;
        printf("shift variant %d: %.3f ms  %.2f T shift-multiplies/s\n", var, ms[var], mu / ms[var] / 1e9);
    }
    {
        std::vector<u64> m0(n), m1(n);
        CHK(hipMemcpy(m0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(m1.data(), d1, n * 8, hipMemcpyDeviceToHost));
        size_t bads = 0, nc = 0;
        for (size_t i = 0; i < n; i++) { if (m0[i] != m1[i]) bads++; if (m1[i] >= GL_P) nc++; }
        printf("gl_shl12 mismatches vs gl_mul_pow2: %zu of %zu; non-canonical: %zu\n", bads, n, nc);
        if (bads || nc) return 1;
    }
    // butterflies again (their outputs are what the check below compares)
    for (int var = 0; var < 2; var++) {
        u64 *o = var ? d1 : d0;
        if (var == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d_in, o);
        else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d_in, o);
        CHK(hipDeviceSynchronize());
    }
    std::vector<u64> r0(n), r1(n);
    CHK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, noncanon = 0;
    for (size_t i = 0; i < n; i++) { if (r0[i] != r1[i]) bad++; if (r1[i] >= GL_P) noncanon++; }
    printf("mismatches vs compiler butterflies: %zu of %zu; non-canonical outputs: %zu\n", bad, n, noncanon);
    return bad || noncanon ? 1 : 0;
}
