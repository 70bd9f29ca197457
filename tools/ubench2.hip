// verification + throughput of candidate field-op formulations (measurement tool, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "/tmp/isa/gl2.hpp"
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, ILP = 8;

__device__ u64 rnd(u64 &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ u64 canon_any(u64 x) { return x >= GL_P ? x - GL_P : x; }

template <int S> __device__ bool chk_shift(u64 a) { return mulpow2<S>(a) == gl_mul(a, gl_pow(2, S)); }

__global__ void verify(unsigned long long *bad) {
    u64 s = 0x1234567ULL + blockIdx.x * 7919 + threadIdx.x * 104729;
    const u64 edge[8] = {0, 1, GL_P - 1, GL_P - 2, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFF00000000ULL, 0x8000000000000000ULL};
    for (int it = 0; it < 4000; it++) {
        u64 a = canon_any(rnd(s)), b = canon_any(rnd(s));
        if (it < 64) { a = edge[it & 7]; b = edge[(it >> 3) & 7]; }
        if (it >= 64 && it < 128) { a = canon_any(rnd(s) & 0xFFFFFFFFULL); b = canon_any(rnd(s) | 0xFFFFFFFF00000000ULL); }
        bool ok = add2(a, b) == gl_add(a, b) && sub2(a, b) == gl_sub(a, b) && mul2(a, b) == gl_mul(a, b);
        ok = ok && chk_shift<12>(a) && chk_shift<24>(a) && chk_shift<36>(a) && chk_shift<48>(a) && chk_shift<60>(a) &&
             chk_shift<72>(a) && chk_shift<84>(a) && chk_shift<32>(a) && chk_shift<64>(a) && chk_shift<3>(a) && chk_shift<93>(a) && chk_shift<31>(a) && chk_shift<63>(a) && chk_shift<65>(a);
        if (!ok) atomicAdd(bad, 1ULL);
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) k(u64 *out, u64 seed) {
    u64 a[ILP];
    u64 b = canon_any(seed + threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x);
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = canon_any(b * (i + 3) + i);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < ILP; i++) {
            if (KIND == 0) a[i] = gl_mul(a[i], b);
            else if (KIND == 1) a[i] = mul2(a[i], b);
            else if (KIND == 2) a[i] = gl_add(a[i], b);
            else if (KIND == 3) a[i] = add2(a[i], b);
            else if (KIND == 4) a[i] = gl_sub(a[i], b);
            else if (KIND == 5) a[i] = sub2(a[i], b);
            else if (KIND == 6) a[i] = mulpow2<12>(a[i]);
            else if (KIND == 7) a[i] = mulpow2<48>(a[i]);
            else if (KIND == 8) a[i] = mulpow2<72>(a[i]);
            else if (KIND == 9) a[i] = mul2(a[i], a[i]);
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s ^= a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int KIND> int run(const char *name) {
    const int blocks = 256 * 8;
    u64 *d; CHK(hipMalloc(&d, blocks * 256 * sizeof(u64)));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345ULL); CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 12345ULL + r);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    double n = 3.0 * blocks * 256 * (double)ITERS * ILP;
    printf("%-24s %8.3f ms  %9.2f Gop/s\n", name, ms, n / ms / 1e6);
    CHK(hipFree(d)); return 0;
}
int main() {
    unsigned long long *bad; CHK(hipMalloc(&bad, 8)); CHK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(verify, dim3(256), dim3(256), 0, 0, bad); CHK(hipDeviceSynchronize());
    unsigned long long hb; CHK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("verify mismatches: %llu\n", hb);
    run<0>("gl_mul"); run<1>("mul2"); run<9>("mul2 (square)"); run<2>("gl_add"); run<3>("add2"); run<4>("gl_sub"); run<5>("sub2");
    run<6>("mulpow2<12>"); run<7>("mulpow2<48>"); run<8>("mulpow2<72>");
    return 0;
}
