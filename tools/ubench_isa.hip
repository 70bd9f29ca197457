// Issue cost of the individual VALU instructions Goldilocks arithmetic is built from, gfx950 (measurement tool).
// Every kernel runs 32 independent instances of ONE instruction per asm block (8 accumulators x 4), 4 waves per SIMD,
// and reports cycles per wave-instruction per SIMD at the clock measured with s_memtime / s_memrealtime.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_isa tools/ubench_isa.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int ITERS = 2048;

#define R4(s) s s s s
// 32-bit accumulators a0..a7 (operands %0..%7), sources %8 %9 (32-bit), SGPR pair temp %10
#define K32(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                              \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                         \
        u32 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u32 b = a0 ^ 0x9E3779B9u, c = a0 * 0x85EBCA6Bu;                                                        \
        u64 st;                                                                                                \
        for (int it = 0; it < ITERS; it++) {                                                                   \
            asm volatile(R4(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 "\n\t")   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                         : "v"(b), "v"(c), "s"(0ULL)                                                           \
                         : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"); \
        }                                                                                                      \
        (void)st;                                                                                              \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                          \
    }
// 64-bit accumulators
#define K64(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                              \
    __global__ void __launch_bounds__(256) NAME(u32 *out, u32 seed) {                                         \
        u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        u64 b = a0 * 0x9E3779B97F4A7C15ULL, c = a0 * 0xC2B2AE3D27D4EB4FULL;                                    \
        for (int it = 0; it < ITERS; it++) {                                                                   \
            asm volatile(R4(I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 "\n\t")   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)       \
                         : "v"(b), "v"(c), "s"(0ULL)                                                           \
                         : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"); \
        }                                                                                                      \
        u64 x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                         \
        out[blockIdx.x * 256 + threadIdx.x] = (u32)x ^ (u32)(x >> 32);                                         \
    }
#define S8(fmt) fmt(0, "s[40:41]"), fmt(1, "s[42:43]"), fmt(2, "s[44:45]"), fmt(3, "s[46:47]"), fmt(4, "s[48:49]"), fmt(5, "s[50:51]"), fmt(6, "s[52:53]"), fmt(7, "s[54:55]")

#define F_ADD(i, s) "v_add_u32 %" #i ", %" #i ", %8"
K32(k_add_u32, F_ADD(0, ), F_ADD(1, ), F_ADD(2, ), F_ADD(3, ), F_ADD(4, ), F_ADD(5, ), F_ADD(6, ), F_ADD(7, ))
#define F_ADDCO(i, s) "v_add_co_u32 %" #i ", " s ", %" #i ", %8"
K32(k_add_co_sgpr, F_ADDCO(0, "s[40:41]"), F_ADDCO(1, "s[42:43]"), F_ADDCO(2, "s[44:45]"), F_ADDCO(3, "s[46:47]"), F_ADDCO(4, "s[48:49]"), F_ADDCO(5, "s[50:51]"), F_ADDCO(6, "s[52:53]"), F_ADDCO(7, "s[54:55]"))
K32(k_add_co_vcc, F_ADDCO(0, "vcc"), F_ADDCO(1, "vcc"), F_ADDCO(2, "vcc"), F_ADDCO(3, "vcc"), F_ADDCO(4, "vcc"), F_ADDCO(5, "vcc"), F_ADDCO(6, "vcc"), F_ADDCO(7, "vcc"))
#define F_ADDC(i, s) "v_addc_co_u32 %" #i ", " s ", %" #i ", %8, " s
K32(k_addc_co_sgpr, F_ADDC(0, "s[40:41]"), F_ADDC(1, "s[42:43]"), F_ADDC(2, "s[44:45]"), F_ADDC(3, "s[46:47]"), F_ADDC(4, "s[48:49]"), F_ADDC(5, "s[50:51]"), F_ADDC(6, "s[52:53]"), F_ADDC(7, "s[54:55]"))
#define F_CND(i, s) "v_cndmask_b32 %" #i ", %" #i ", %8, " s
K32(k_cndmask_sgpr, F_CND(0, "s[40:41]"), F_CND(1, "s[42:43]"), F_CND(2, "s[44:45]"), F_CND(3, "s[46:47]"), F_CND(4, "s[48:49]"), F_CND(5, "s[50:51]"), F_CND(6, "s[52:53]"), F_CND(7, "s[54:55]"))
K32(k_cndmask_vcc, F_CND(0, "vcc"), F_CND(1, "vcc"), F_CND(2, "vcc"), F_CND(3, "vcc"), F_CND(4, "vcc"), F_CND(5, "vcc"), F_CND(6, "vcc"), F_CND(7, "vcc"))
#define F_ALIGN(i, s) "v_alignbit_b32 %" #i ", %" #i ", %8, 13"
K32(k_alignbit, F_ALIGN(0, ), F_ALIGN(1, ), F_ALIGN(2, ), F_ALIGN(3, ), F_ALIGN(4, ), F_ALIGN(5, ), F_ALIGN(6, ), F_ALIGN(7, ))
#define F_ADD3(i, s) "v_add3_u32 %" #i ", %" #i ", %8, %9"
K32(k_add3, F_ADD3(0, ), F_ADD3(1, ), F_ADD3(2, ), F_ADD3(3, ), F_ADD3(4, ), F_ADD3(5, ), F_ADD3(6, ), F_ADD3(7, ))
#define F_BITOP(i, s) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xe8"
K32(k_bitop3, F_BITOP(0, ), F_BITOP(1, ), F_BITOP(2, ), F_BITOP(3, ), F_BITOP(4, ), F_BITOP(5, ), F_BITOP(6, ), F_BITOP(7, ))
#define F_MULLO(i, s) "v_mul_lo_u32 %" #i ", %" #i ", %8"
K32(k_mul_lo, F_MULLO(0, ), F_MULLO(1, ), F_MULLO(2, ), F_MULLO(3, ), F_MULLO(4, ), F_MULLO(5, ), F_MULLO(6, ), F_MULLO(7, ))
#define F_MULHI(i, s) "v_mul_hi_u32 %" #i ", %" #i ", %8"
K32(k_mul_hi, F_MULHI(0, ), F_MULHI(1, ), F_MULHI(2, ), F_MULHI(3, ), F_MULHI(4, ), F_MULHI(5, ), F_MULHI(6, ), F_MULHI(7, ))
#define F_CMP32(i, s) "v_cmp_lt_u32 " s ", %" #i ", %8"
K32(k_cmp_u32_sgpr, F_CMP32(0, "s[40:41]"), F_CMP32(1, "s[42:43]"), F_CMP32(2, "s[44:45]"), F_CMP32(3, "s[46:47]"), F_CMP32(4, "s[48:49]"), F_CMP32(5, "s[50:51]"), F_CMP32(6, "s[52:53]"), F_CMP32(7, "s[54:55]"))
#define F_MIN(i, s) "v_min_u32 %" #i ", %" #i ", %8"
K32(k_min_u32, F_MIN(0, ), F_MIN(1, ), F_MIN(2, ), F_MIN(3, ), F_MIN(4, ), F_MIN(5, ), F_MIN(6, ), F_MIN(7, ))
#define F_MOV(i, s) "v_mov_b32 %" #i ", %8"
K32(k_mov_b32, F_MOV(0, ), F_MOV(1, ), F_MOV(2, ), F_MOV(3, ), F_MOV(4, ), F_MOV(5, ), F_MOV(6, ), F_MOV(7, ))
#define F_LSHLADD32(i, s) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8"
K32(k_lshl_add_u32, F_LSHLADD32(0, ), F_LSHLADD32(1, ), F_LSHLADD32(2, ), F_LSHLADD32(3, ), F_LSHLADD32(4, ), F_LSHLADD32(5, ), F_LSHLADD32(6, ), F_LSHLADD32(7, ))
#define F_ASHR(i, s) "v_ashrrev_i32 %" #i ", 31, %" #i
K32(k_ashr, F_ASHR(0, ), F_ASHR(1, ), F_ASHR(2, ), F_ASHR(3, ), F_ASHR(4, ), F_ASHR(5, ), F_ASHR(6, ), F_ASHR(7, ))

#define F_LA64(i, s) "v_lshl_add_u64 %" #i ", %" #i ", 0, %8"
K64(k_lshl_add_u64, F_LA64(0, ), F_LA64(1, ), F_LA64(2, ), F_LA64(3, ), F_LA64(4, ), F_LA64(5, ), F_LA64(6, ), F_LA64(7, ))
#define F_CMP64(i, s) "v_cmp_lt_u64 " s ", %" #i ", %8"
K64(k_cmp_u64_sgpr, F_CMP64(0, "s[40:41]"), F_CMP64(1, "s[42:43]"), F_CMP64(2, "s[44:45]"), F_CMP64(3, "s[46:47]"), F_CMP64(4, "s[48:49]"), F_CMP64(5, "s[50:51]"), F_CMP64(6, "s[52:53]"), F_CMP64(7, "s[54:55]"))
K64(k_cmp_u64_vcc, F_CMP64(0, "vcc"), F_CMP64(1, "vcc"), F_CMP64(2, "vcc"), F_CMP64(3, "vcc"), F_CMP64(4, "vcc"), F_CMP64(5, "vcc"), F_CMP64(6, "vcc"), F_CMP64(7, "vcc"))
#define F_SHL64(i, s) "v_lshlrev_b64 %" #i ", 5, %" #i
K64(k_lshlrev_b64, F_SHL64(0, ), F_SHL64(1, ), F_SHL64(2, ), F_SHL64(3, ), F_SHL64(4, ), F_SHL64(5, ), F_SHL64(6, ), F_SHL64(7, ))
#define F_MOV64(i, s) "v_mov_b64 %" #i ", %8"
K64(k_mov_b64, F_MOV64(0, ), F_MOV64(1, ), F_MOV64(2, ), F_MOV64(3, ), F_MOV64(4, ), F_MOV64(5, ), F_MOV64(6, ), F_MOV64(7, ))
// v_mad_u64_u32 vdst(64), sdst, src0, src1, src2(64): acc = lo(acc)*b + acc  (sources: sub-registers not addressable -> use %8 %9)
#define F_MAD(i, s) "v_mad_u64_u32 %" #i ", " s ", %8, %9, %" #i
__global__ void __launch_bounds__(256) k_mad_u64_u32(u32 *out, u32 seed) {
    u64 a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    u32 b = (u32)a0 ^ 0x9E3779B9u, c = (u32)a0 * 0x85EBCA6Bu;
    for (int it = 0; it < ITERS; it++) {
        asm volatile(R4(F_MAD(0, "s[40:41]") "\n\t" F_MAD(1, "s[42:43]") "\n\t" F_MAD(2, "s[44:45]") "\n\t" F_MAD(3, "s[46:47]") "\n\t"
                        F_MAD(4, "s[48:49]") "\n\t" F_MAD(5, "s[50:51]") "\n\t" F_MAD(6, "s[52:53]") "\n\t" F_MAD(7, "s[54:55]") "\n\t")
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(b), "v"(c)
                     : "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
    }
    u64 x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    out[blockIdx.x * 256 + threadIdx.x] = (u32)x ^ (u32)(x >> 32);
}
#define F_MAD24(i, s) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
K32(k_mad_u32_u24, F_MAD24(0, ), F_MAD24(1, ), F_MAD24(2, ), F_MAD24(3, ), F_MAD24(4, ), F_MAD24(5, ), F_MAD24(6, ), F_MAD24(7, ))
#define F_MUL24(i, s) "v_mul_u32_u24 %" #i ", %" #i ", %8"
K32(k_mul_u32_u24, F_MUL24(0, ), F_MUL24(1, ), F_MUL24(2, ), F_MUL24(3, ), F_MUL24(4, ), F_MUL24(5, ), F_MUL24(6, ), F_MUL24(7, ))
#define F_DOT4(i, s) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i
K32(k_dot4_u32_u8, F_DOT4(0, ), F_DOT4(1, ), F_DOT4(2, ), F_DOT4(3, ), F_DOT4(4, ), F_DOT4(5, ), F_DOT4(6, ), F_DOT4(7, ))
#define F_PERM(i, s) "v_perm_b32 %" #i ", %" #i ", %8, %9"
K32(k_perm_b32, F_PERM(0, ), F_PERM(1, ), F_PERM(2, ), F_PERM(3, ), F_PERM(4, ), F_PERM(5, ), F_PERM(6, ), F_PERM(7, ))
#define F_PKADD(i, s) "v_pk_add_u16 %" #i ", %" #i ", %8"
K32(k_pk_add_u16, F_PKADD(0, ), F_PKADD(1, ), F_PKADD(2, ), F_PKADD(3, ), F_PKADD(4, ), F_PKADD(5, ), F_PKADD(6, ), F_PKADD(7, ))
#define F_PKMAD(i, s) "v_pk_mad_u16 %" #i ", %" #i ", %8, %9"
K32(k_pk_mad_u16, F_PKMAD(0, ), F_PKMAD(1, ), F_PKMAD(2, ), F_PKMAD(3, ), F_PKMAD(4, ), F_PKMAD(5, ), F_PKMAD(6, ), F_PKMAD(7, ))
#define F_BFE(i, s) "v_bfe_u32 %" #i ", %" #i ", 3, 29"
K32(k_bfe_u32, F_BFE(0, ), F_BFE(1, ), F_BFE(2, ), F_BFE(3, ), F_BFE(4, ), F_BFE(5, ), F_BFE(6, ), F_BFE(7, ))
#define F_FMA64(i, s) "v_fma_f64 %" #i ", %" #i ", 1.0, 0"
K64(k_fma_f64, F_FMA64(0, ), F_FMA64(1, ), F_FMA64(2, ), F_FMA64(3, ), F_FMA64(4, ), F_FMA64(5, ), F_FMA64(6, ), F_FMA64(7, ))
#define F_PKFMA32(i, s) "v_pk_fma_f32 %" #i ", %" #i ", 1.0, 0"
K64(k_pk_fma_f32, F_PKFMA32(0, ), F_PKFMA32(1, ), F_PKFMA32(2, ), F_PKFMA32(3, ), F_PKFMA32(4, ), F_PKFMA32(5, ), F_PKFMA32(6, ), F_PKFMA32(7, ))
// clock probe: cycles per 100 MHz tick
__global__ void k_clock(u64 *out) {
    u64 t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    u32 a = threadIdx.x;
    for (int i = 0; i < 200000; i++) asm volatile("v_add_u32 %0, %0, %0" : "+v"(a));
    u64 t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}

typedef void (*kern_t)(u32 *, u32);
static double run(const char *name, kern_t k, u32 *d, int waves_per_simd, double ghz) {
    const int blocks = 256 * waves_per_simd;   // 256-thread blocks = 4 waves = 1 wave per SIMD each
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 1u);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, 2u);
    CHK(hipEventRecord(e1)); CHK(hipDeviceSynchronize());
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)ITERS * 32 * waves_per_simd;
    const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
    printf("%-18s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, waves_per_simd, ms, cyc, ghz);
    return cyc;
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
        printf("clock probe: %llu shader cycles in %llu x 10 ns -> %.3f GHz\n", h[0], h[1], ghz);
    }
#define RUN(k) for (int w : {1, 2, 4, 8}) run(#k, k, d, w, ghz)
    RUN(k_add_u32); RUN(k_add_co_sgpr); RUN(k_add_co_vcc); RUN(k_addc_co_sgpr); RUN(k_cndmask_sgpr); RUN(k_cndmask_vcc);
    RUN(k_alignbit); RUN(k_add3); RUN(k_bitop3); RUN(k_mul_lo); RUN(k_mul_hi); RUN(k_cmp_u32_sgpr); RUN(k_min_u32);
    RUN(k_mov_b32); RUN(k_lshl_add_u32); RUN(k_ashr);
    RUN(k_lshl_add_u64); RUN(k_cmp_u64_sgpr); RUN(k_cmp_u64_vcc); RUN(k_lshlrev_b64); RUN(k_mov_b64); RUN(k_mad_u64_u32);
    RUN(k_mad_u32_u24); RUN(k_mul_u32_u24); RUN(k_dot4_u32_u8); RUN(k_perm_b32); RUN(k_pk_add_u16); RUN(k_pk_mad_u16); RUN(k_bfe_u32); RUN(k_fma_f64); RUN(k_pk_fma_f32);
    return 0;
}
