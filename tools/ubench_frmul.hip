// Latency of ONE Montgomery product of F_r (csrc/fr254.hpp: nine 29-bit limbs) on a wave that has its SIMD to itself -- the situation of the
// Poseidon-BN254 kernels (the cooperative ones walk a chain of dependent products; the bulk one runs one wave per SIMD): fr_mul as shipped
// (ONE column accumulator: every multiply-add of a column waits for the one before it) against a form with the column's terms split over
// two accumulators.  Measurement tool.  build + run: hipcc -O3 --offload-arch=gfx950 -I eigen_zeth_amd/csrc tools/ubench_frmul.hip -o /tmp/ubench_frmul && /tmp/ubench_frmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include "fr254.hpp"

__device__ __forceinline__ fr fr_mul_2acc(const fr &a, const fr &b) {
    u32 m[9], t[9];
    u64 acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        u64 a0 = acc, a1 = 0;
#pragma unroll
        for (int i = 0; i <= k; i++) { if (i & 1) a1 += (u64)a.l[i] * b.l[k - i]; else a0 += (u64)a.l[i] * b.l[k - i]; }
#pragma unroll
        for (int i = 0; i < k; i++) { if (i & 1) a0 += (u64)m[i] * fr_p(k - i); else a1 += (u64)m[i] * fr_p(k - i); }
        acc = a0 + a1;
        m[k] = ((u32)acc * FR_INV29) & FR_MASK;
        acc += (u64)m[k] * fr_p(0);
        acc >>= FR_B;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        u64 a0 = acc, a1 = 0;
#pragma unroll
        for (int i = k - 8; i < 9; i++) {
            a0 += (u64)a.l[i] * b.l[k - i];
            a1 += (u64)m[i] * fr_p(k - i);
        }
        acc = a0 + a1;
        t[k - 9] = (u32)acc & FR_MASK;
        acc >>= FR_B;
    }
    t[8] = (u32)acc;
    return fr_norm_sub(t);
}

template <int MODE>
__global__ void __launch_bounds__(64) chain(u32 *out, int n) {
    fr x;
    for (int i = 0; i < 9; i++) x.l[i] = (threadIdx.x * 977u + i * 31u + 5u) & FR_MASK;
    x.l[8] &= 0x1FFFFF;
    for (int it = 0; it < n; it++) x = MODE == 0 ? fr_mul(x, x) : fr_mul_2acc(x, x);
    for (int i = 0; i < 9; i++) out[(blockIdx.x * 64 + threadIdx.x) * 9 + i] = x.l[i];
}

int main() {
    u32 *d;
    hipMalloc(&d, 1024 * 64 * 9 * 4 * 8);
    const int n = 20000;
    for (int waves_per_simd : {1, 2, 4}) {
        const int blocks = 256 * 4 * waves_per_simd;
        for (int mode = 0; mode < 2; mode++) {
            double best = 1e9;
            u32 h0[9];
            for (int rep = 0; rep < 3; rep++) {
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(blocks), dim3(64), 0, 0, d, n);
                else hipLaunchKernelGGL(chain<1>, dim3(blocks), dim3(64), 0, 0, d, n);
                hipDeviceSynchronize();
                const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (s < best) best = s;
            }
            hipMemcpy(h0, d, 36, hipMemcpyDeviceToHost);
            printf("%d wave(s) per SIMD, %s: %.1f ns per dependent product per wave (%.0f cycles at 2.4 GHz); chip %.2f G products/s; check %08x\n", waves_per_simd,
                   mode == 0 ? "one accumulator (fr_mul)      " : "two accumulators per column   ", best / n * 1e9, best / n * 2.4e9,
                   (double)blocks * 64 * n / best / 1e9, h0[0] ^ h0[3] ^ h0[8]);
        }
    }
    return 0;
}
