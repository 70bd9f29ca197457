// HBM access-pattern microbenchmarks for the NTT pass design on gfx950 (measurement tool, not product).
//   copy      : contiguous 16 B/lane copy (the achievable ceiling; plain and non-temporal)
//   runs      : every workgroup moves an R x T tile of u64: R runs of T*8 bytes at a column stride of N/R elements,
//               written back either as one contiguous block (Stockham pass 1) or as runs again (pass 2)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_mem tools/ubench_mem.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
typedef __attribute__((ext_vector_type(2))) unsigned long long u64x2;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ u64x2 ld16(const u64x2 *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ void st16(u64x2 *p, u64x2 v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}

template <bool NT>
__global__ void __launch_bounds__(256) copy_kernel(const u64x2 *__restrict__ in, u64x2 *__restrict__ out, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u64x2 a = ld16<NT>(in + i), b = ld16<NT>(in + i + stride), c = ld16<NT>(in + i + 2 * stride), d = ld16<NT>(in + i + 3 * stride);
        st16<NT>(out + i, a); st16<NT>(out + i + stride, b); st16<NT>(out + i + 2 * stride, c); st16<NT>(out + i + 3 * stride, d);
    }
    for (; i < n16; i += stride) st16<NT>(out + i, ld16<NT>(in + i));
}

// tile of R rows x T elements (T*8 = run bytes); NTH threads, each moves E = R*T/NTH elements as E/2 16-byte pieces.
// WMODE 0: contiguous block out[tile*R*T ...]; WMODE 1: same run pattern as the reads.
// XMAP 1: tiles that are adjacent in u (sharing 128-byte lines when T*8 < 128) go to workgroups with equal blockIdx%8
template <int LOGR, int LOGT, int NTH, int WMODE, int XMAP, bool NT>
__global__ void __launch_bounds__(NTH) runs_kernel(const u64 *__restrict__ in, u64 *__restrict__ out, int logn, int tiles_per_col_log) {
    constexpr int R = 1 << LOGR, T = 1 << LOGT;
    constexpr int P = R * T / 2;          // 16-byte pieces per tile
    constexpr int PPT = P / NTH;          // pieces per thread
    constexpr int PPR = T / 2;            // pieces per run
    static_assert(PPT >= 1 && T >= 2, "shape");
    const u64 N = 1ULL << logn;
    unsigned b = blockIdx.x;
    const unsigned ntiles = 1u << tiles_per_col_log;
    unsigned tile;
    if (XMAP) {
        // group of 8 consecutive blockIdx = 8 XCDs; XCD x walks tiles [x*ntiles/8, (x+1)*ntiles/8)
        tile = (b & 7u) * (ntiles >> 3) + (b >> 3);
    } else tile = b;
    const u64 col = blockIdx.y;
    const u64 *src = in + col * N;
    u64 *dst = out + col * N;
    const u64 u0 = (u64)tile << LOGT;
    const int logNR = logn - LOGR;
    u64x2 v[PPT];
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int p = i * NTH + threadIdx.x;
        const int r = p / PPR, q = p % PPR;
        v[i] = ld16<NT>((const u64x2 *)(src + ((u64)r << logNR) + u0 + 2 * q));
    }
#pragma unroll
    for (int i = 0; i < PPT; i++) {
        const int p = i * NTH + threadIdx.x;
        if (WMODE == 0) {
            st16<NT>((u64x2 *)(dst + (u0 << LOGR) + 2 * (u64)p), v[i]);
        } else {
            const int r = p / PPR, q = p % PPR;
            st16<NT>((u64x2 *)(dst + ((u64)r << logNR) + u0 + 2 * q), v[i]);
        }
    }
}

static float time_it(void (*launch)(void *), void *arg, int reps) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    launch(arg); launch(arg);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch(arg);
    CHK(hipEventRecord(e1));
    CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

struct Bufs { u64 *in, *out; size_t n; int logn, cols; };

template <bool NT> static void l_copy(void *a) {
    Bufs *b = (Bufs *)a;
    hipLaunchKernelGGL(copy_kernel<NT>, dim3(256 * 8), dim3(256), 0, 0, (const u64x2 *)b->in, (u64x2 *)b->out, b->n / 2);
}
template <int LOGR, int LOGT, int NTH, int WMODE, int XMAP, bool NT> static void l_runs(void *a) {
    Bufs *b = (Bufs *)a;
    const int tl = b->logn - LOGR - LOGT;
    hipLaunchKernelGGL((runs_kernel<LOGR, LOGT, NTH, WMODE, XMAP, NT>), dim3(1u << tl, b->cols), dim3(NTH), 0, 0, b->in, b->out, b->logn, tl);
}

static void report(const char *name, float ms, const Bufs &b) {
    const double bytes = 2.0 * b.n * 8;
    printf("%-58s %8.3f ms  %8.1f GB/s (read+write)\n", name, ms, bytes / ms / 1e6);
    fflush(stdout);
}

#define RUNS(LOGR, LOGT, NTH, WMODE, XMAP, NT) \
    report("runs R=2^" #LOGR " T=2^" #LOGT " nth=" #NTH " wmode=" #WMODE " xmap=" #XMAP " nt=" #NT, \
           time_it(l_runs<LOGR, LOGT, NTH, WMODE, XMAP, NT>, &b, 5), b)

int main(int argc, char **argv) {
    Bufs b;
    b.logn = 24; b.cols = 16;
    b.n = (size_t)b.cols << b.logn;
    CHK(hipMalloc(&b.in, b.n * 8)); CHK(hipMalloc(&b.out, b.n * 8));
    CHK(hipMemset(b.in, 1, b.n * 8)); CHK(hipMemset(b.out, 0, b.n * 8));
    report("copy 16B/lane plain", time_it(l_copy<false>, &b, 5), b);
    report("copy 16B/lane nontemporal", time_it(l_copy<true>, &b, 5), b);
    // radix-256 shapes (the round-1 kernel: T=16, 256 threads)
    RUNS(8, 4, 256, 0, 0, true);
    RUNS(8, 4, 256, 1, 0, true);
    RUNS(8, 5, 512, 0, 0, true);
    RUNS(8, 5, 512, 1, 0, true);
    RUNS(8, 4, 256, 1, 0, false);
    // radix-4096 shapes (two-pass plan at 2^24): T=4 (32-byte runs) and T=8 (64-byte runs, does not fit LDS: bandwidth only)
    RUNS(12, 2, 1024, 0, 0, true);
    RUNS(12, 2, 1024, 0, 1, true);
    RUNS(12, 2, 1024, 1, 0, true);
    RUNS(12, 2, 1024, 1, 1, true);
    RUNS(12, 2, 1024, 0, 0, false);
    RUNS(12, 2, 1024, 0, 1, false);
    RUNS(12, 2, 1024, 1, 0, false);
    RUNS(12, 2, 1024, 1, 1, false);
    RUNS(12, 2, 256, 1, 1, false);
    RUNS(12, 2, 256, 1, 1, true);
    RUNS(12, 3, 1024, 0, 0, true);
    RUNS(12, 3, 1024, 1, 0, true);
    RUNS(12, 3, 1024, 0, 1, false);
    RUNS(12, 3, 1024, 1, 1, false);
    // radix-2^10 / 2^11 shapes
    RUNS(10, 4, 1024, 0, 0, true);
    RUNS(10, 4, 1024, 1, 0, true);
    RUNS(11, 3, 1024, 0, 0, true);
    RUNS(11, 3, 1024, 1, 0, true);
    RUNS(11, 3, 1024, 1, 1, false);
    return 0;
}
