// Instruction-throughput microbenchmarks for the integer paths that bound Goldilocks arithmetic on
// gfx950: run on the GPU box, prints ops/s per variant.  (measurement tool, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../eigen_zeth_amd/csrc/gl.hpp"

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int ILP = 8;

template <int KIND>
__global__ void __launch_bounds__(256) k(u64 *out, u64 seed) {
    u64 a[ILP];
    u64 b = seed + threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x;
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = b * (i + 3) + i;
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (KIND == 0) a[i] = gl_mul(a[i], b);                       // full modmul
            else if (KIND == 1) a[i] = gl_add(a[i], b);                  // modadd
            else if (KIND == 2) a[i] = (u64)(u32)a[i] * (u32)b + a[i];   // v_mad_u64_u32
            else if (KIND == 3) { u32 x = (u32)a[i]; x = x * (u32)b + 7u; a[i] = x; }  // v_mul_lo_u32 (+add)
            else if (KIND == 4) { u32 x = (u32)a[i]; x = __umulhi(x, (u32)b) + 1u; a[i] = x; }  // v_mul_hi_u32
            else if (KIND == 5) { u32 x = (u32)a[i]; x = __umul24(x, (u32)b) + x; a[i] = x; }   // v_mad_u32_u24
            else if (KIND == 6) a[i] = (a[i] << 13) ^ (a[i] >> 7);       // 64-bit shifts
            else if (KIND == 7) a[i] = a[i] + b;                          // 64-bit add
            else if (KIND == 8) a[i] = gl_sub(a[i], b);
            else if (KIND == 9) { u32 x = (u32)a[i]; x = x + (u32)b; x ^= x >> 3; a[i] = x; }   // 32-bit add+shift+xor (3 ops)
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
int run(const char *name, double ops_per_iter) {
    const int blocks = 256 * 8;
    u64 *d;
    CHK(hipMalloc(&d, blocks * 256 * sizeof(u64)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345ULL);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345ULL + r);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double n = 3.0 * blocks * 256 * (double)ITERS * ILP;
    printf("%-28s %8.3f ms  %10.2f Gop/s\n", name, ms, n / ms / 1e6);
    CHK(hipFree(d));
    return 0;
}

int main() {
    run<0>("gl_mul (modmul)", 1);
    run<1>("gl_add (modadd)", 1);
    run<8>("gl_sub (modsub)", 1);
    run<2>("v_mad_u64_u32", 1);
    run<3>("v_mul_lo_u32+add", 1);
    run<4>("v_mul_hi_u32+add", 1);
    run<5>("v_mad_u32_u24", 1);
    run<6>("shl64/shr64/xor", 1);
    run<7>("add64", 1);
    run<9>("add32+shr+xor", 1);
    return 0;
}
