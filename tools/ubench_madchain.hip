// Latency of a DEPENDENT v_mad_u64_u32 accumulation chain on gfx950 (measurement tool): the column accumulators of the 29-bit-limb field
// products (csrc/fr254.hpp, csrc/msm.hip) are chains  acc = a * b + acc  on ONE register pair.  32 mads per asm block spread over 1, 2, 4 or 8
// accumulators, 1 / 2 / 4 waves per SIMD: cycles per wave-instruction per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_madchain tools/ubench_madchain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048;
#define F_MAD(i) "v_mad_u64_u32 %" #i ", s[40:41], %8, %9, %" #i "\n\t"
#define R4(s) s s s s
#define KERN(NAME, BODY)                                                                                        \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                          \
        u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u32 b = (u32)a0 ^ 0x9E3779B9u, c = (u32)a0 * 0x85EBCA6Bu;                                               \
        for (int it = 0; it < ITERS; it++) {                                                                    \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                         : "v"(b), "v"(c) : "vcc", "scc", "s40", "s41");                                        \
        }                                                                                                       \
        u64 x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (u32)x ^ (u32)(x >> 32);                                          \
    }
KERN(k_chain1, R4(R4(F_MAD(0) F_MAD(0))))
KERN(k_chain2, R4(R4(F_MAD(0) F_MAD(1))))
KERN(k_chain4, R4(F_MAD(0) F_MAD(1) F_MAD(2) F_MAD(3) F_MAD(0) F_MAD(1) F_MAD(2) F_MAD(3)))
KERN(k_chain8, R4(F_MAD(0) F_MAD(1) F_MAD(2) F_MAD(3) F_MAD(4) F_MAD(5) F_MAD(6) F_MAD(7)))

__global__ void k_clock(u64 *out) {
    u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32 a = threadIdx.x;
    for (int i = 0; i < 200000; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}
typedef void (*kern_t)(u32 *, u32);
static void run(const char *name, kern_t k, u32 *d, int wps, double ghz) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2u);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-10s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, wps, ms, ms * 1e-3 * ghz * 1e9 / ((double)ITERS * 32 * wps), ghz);
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    u32 *d; CHK(hipMalloc(&d, 256 * 8 * 256 * 4));
    u64 *dc; CHK(hipMalloc(&dc, 64));
    double ghz = 2.1;
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(k_clock, dim3(1024), dim3(256), 0, 0, dc);
        u64 h[3]; CHK(hipMemcpy(h, dc, 24, hipMemcpyDeviceToHost));
        ghz = (double)h[0] / (double)h[1] * 0.1;
    }
    printf("clock %.3f GHz\n", ghz);
#define RUN(k) for (int w : {1, 2, 4}) run(#k, k, d, w, ghz)
    RUN(k_chain1); RUN(k_chain2); RUN(k_chain4); RUN(k_chain8);
    return 0;
}
