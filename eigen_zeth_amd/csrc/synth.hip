// Synthetic witnesses generated IN HBM (gfx950): the generators of zp_synth_trace_bound (csrc/capi.hip) as kernels, word for word the same
// traces.  They stand in for the zkVM executor (proto/prover/v1/prover.proto:49-54, 80-91; not obtainable offline) where the host
// generator bounds a batch: 64 chunks of 2^22 rows x 76 columns are 60 CPU-seconds of a sequential degree-2 recurrence, 5 s on 12
// threads against 3.5 s of proving (bench.py batch_proof.config5_one_gpu).
//
// The wide-mix columns are a recurrence over the ROWS (row r+1 = f(row r): the AIR's transition constraint), so one trace cannot be
// filled in parallel from its first row.  Two kernels: mix_checkpoint_kernel walks the recurrence of a whole BATCH of chunks, one
// wave per chunk (the chunks are independent), keeping the state at every SEG-th row (0.5 MB per chunk); mix_expand_kernel then
// fills a trace from its checkpoints, one wave per segment: N / SEG independent walks of SEG rows.  A walk keeps the row in
// registers (ceil(Ww / 64) columns per lane) and meets its neighbours c+1, c+2 through LDS.  The other columns are closed forms:
// Fibonacci from the fast-doubling identities per block of rows, the range values straight from the counter-based generator
// (splitmix64: draw k is a function of seed + k), the multiplicities by atomics.
#include <hip/hip_runtime.h>

#include <vector>

#include "ctx.hpp"
#include "gl.hpp"
#include "gl_asm.hpp"

namespace {

constexpr int SEG_LOG = 12;                 // rows per checkpoint segment
constexpr int MAX_CPL = 4;                  // wide-mix columns per lane: Ww <= 256

__host__ __device__ inline u64 splitmix_draw(u64 seed0, u64 k) {   // the k-th value (k = 0, 1, ..) of capi.hip's `next`
    u64 z = seed0 + (k + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z >= GL_P ? z - GL_P : z;
}

// one row step of the wide mix for the columns this lane holds: c_i <- c_i c_(i+1) + c_(i+2) + i  (indices mod Ww)
template <int CPL>
__device__ __forceinline__ void mix_step(u64 *cur, u64 *buf, int lane, int Ww) {
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int col = lane + 64 * c;
        if (col < Ww) {
            buf[col] = cur[c];
            if (col < 2) buf[Ww + col] = cur[c];      // wrap-around neighbours without a modulo
        }
    }
    __syncthreads();                                    // one wave: orders its LDS writes before its reads
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int col = lane + 64 * c;
        if (col < Ww) cur[c] = gl_add(gl_add(gl_mul1(cur[c], buf[col + 1]), buf[col + 2]), (u64)col);   // gl_asm.hpp: the 17-instruction product
    }
}

// init: u64[chunk][Ww] first rows;  ckpt: u64[chunk][N >> seg_log][Ww], the row at every multiple of 2^seg_log
template <int CPL>
__global__ void __launch_bounds__(64) mix_checkpoint_kernel(const u64 *__restrict__ init, u64 *__restrict__ ckpt, int logn, int seg_log, int Ww) {
    __shared__ u64 lds[2][64 * MAX_CPL + 2];
    const int lane = threadIdx.x;
    const size_t chunk = blockIdx.x, N = (size_t)1 << logn, nseg = N >> seg_log;
    u64 cur[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) cur[c] = lane + 64 * c < Ww ? init[chunk * Ww + lane + 64 * c] : 0;
    u64 *out = ckpt + chunk * nseg * Ww;
    for (size_t s = 0; s < nseg; s++) {
#pragma unroll
        for (int c = 0; c < CPL; c++)
            if (lane + 64 * c < Ww) out[s * Ww + lane + 64 * c] = cur[c];
        if (s + 1 == nseg) break;
        for (int r = 0; r < (1 << seg_log); r++) mix_step<CPL>(cur, lds[r & 1], lane, Ww);
    }
}
// trace columns [0, Ww) of one chunk from its checkpoints: block = segment
template <int CPL>
__global__ void __launch_bounds__(64) mix_expand_kernel(const u64 *__restrict__ ckpt, u64 *__restrict__ trace, int logn, int seg_log, int Ww) {
    __shared__ u64 lds[2][64 * MAX_CPL + 2];
    const int lane = threadIdx.x;
    const size_t s = blockIdx.x, N = (size_t)1 << logn, r0 = s << seg_log;
    u64 cur[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) cur[c] = lane + 64 * c < Ww ? ckpt[s * Ww + lane + 64 * c] : 0;
    for (int r = 0; r < (1 << seg_log); r++) {
#pragma unroll
        for (int c = 0; c < CPL; c++)
            if (lane + 64 * c < Ww) trace[(size_t)(lane + 64 * c) * N + r0 + r] = cur[c];
        mix_step<CPL>(cur, lds[r & 1], lane, Ww);
    }
}

// The same with the rows STAGED through LDS (round 5): the kernel above stores row by row -- 64 lanes, 64 columns, 8 bytes each, 2^logn rows apart --
// and a 2^22 x 64 trace is 2 GiB of such stores.  Here a block of RB = 64 / CPL rows is collected in LDS (row stride Ww + 1 words: no bank
// conflicts either way) and written out column by column: RB consecutive rows of one column per RB lanes, 512-byte runs for one column per lane.
template <int CPL>
__global__ void __launch_bounds__(64) mix_expand_tiled_kernel(const u64 *__restrict__ ckpt, u64 *__restrict__ trace, int logn, int seg_log, int Ww) {
    constexpr int RB = CPL == 1 ? 64 : CPL == 2 ? 32 : 16;
    __shared__ u64 lds[2][64 * MAX_CPL + 2];
    __shared__ u64 tile[RB * (64 * CPL + 1)];
    const int lane = threadIdx.x, stride = Ww + 1;
    const size_t s = blockIdx.x, N = (size_t)1 << logn, r0 = s << seg_log;
    u64 cur[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) cur[c] = lane + 64 * c < Ww ? ckpt[s * Ww + lane + 64 * c] : 0;
    const int rows = 1 << seg_log;
    for (int rb = 0; rb < rows; rb += RB) {
        const int nr = rows - rb < RB ? rows - rb : RB;         // (a segment shorter than RB: logn < log2 RB)
        for (int r = 0; r < nr; r++) {
#pragma unroll
            for (int c = 0; c < CPL; c++)
                if (lane + 64 * c < Ww) tile[r * stride + lane + 64 * c] = cur[c];
            mix_step<CPL>(cur, lds[r & 1], lane, Ww);
        }
        __syncthreads();
        for (int i = lane; i < Ww * RB; i += 64) {
            const int col = i / RB, rr = i % RB;
            if (rr < nr) trace[(size_t)col * N + r0 + rb + rr] = tile[rr * stride + col];
        }
        __syncthreads();
    }
}

// (F(k), F(k + 1)) mod p by fast doubling
__host__ __device__ inline void fib_pair(u64 k, u64 &fk, u64 &fk1) {
    u64 a = 0, b = 1;                           // F(0), F(1)
    for (int bit = 63; bit >= 0; bit--) {
        const u64 t = gl_sub(gl_add(b, b), a);  // 2 F(n+1) - F(n)
        const u64 c = gl_mul(a, t);             // F(2n)
        const u64 d = gl_add(gl_mul(a, a), gl_mul(b, b));   // F(2n+1)
        a = c;
        b = d;
        if ((k >> bit) & 1) {
            const u64 e = gl_add(a, b);
            a = b;
            b = e;
        }
    }
    fk = a;
    fk1 = b;
}
// the pair (a, b) of the recurrence (a, b) <- (b, a + b) at row k:  a_k = F(k-1) a_0 + F(k) b_0,  b_k = F(k) a_0 + F(k+1) b_0
__host__ __device__ inline void fib_at(u64 a0, u64 b0, u64 k, u64 &ak, u64 &bk) {
    if (k == 0) {
        ak = a0;
        bk = b0;
        return;
    }
    u64 f0, f1;                                 // F(k-1), F(k)
    fib_pair(k - 1, f0, f1);
    const u64 f2 = gl_add(f0, f1);
    ak = gl_add(gl_mul(f0, a0), gl_mul(f1, b0));
    bk = gl_add(gl_mul(f1, a0), gl_mul(f2, b0));
}
constexpr int FIB_BLOCK = 256;                  // rows per thread
__global__ void __launch_bounds__(256) fib_kernel(u64 *__restrict__ fa, u64 *__restrict__ fb, size_t N, u64 a0, u64 b0) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, r0 = t * FIB_BLOCK;
    if (r0 >= N) return;
    u64 a, b;
    fib_at(a0, b0, r0, a, b);
    for (size_t r = r0; r < r0 + FIB_BLOCK && r < N; r++) {
        fa[r] = a;
        fb[r] = b;
        const u64 e = gl_add(a, b);
        a = b;
        b = e;
    }
}

// chunk AIR, the six columns behind the Fibonacci pair: r (draw base + i, k bits), r permuted, table, multiplicities (+= 1 at r << rep), r^2, a r + b
__global__ void __launch_bounds__(256) range_kernel(u64 *__restrict__ rv, u64 *__restrict__ qv, u64 *__restrict__ tv, u64 *__restrict__ mv, u64 *__restrict__ cv,
                                                    u64 *__restrict__ dv, const u64 *__restrict__ fa, const u64 *__restrict__ fb, size_t N, u64 seed0, u64 base,
                                                    int k, int rep) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const u64 mask = ((u64)1 << k) - 1;
    const u64 r = splitmix_draw(seed0, base + i) & mask;
    rv[i] = r;
    qv[i] = splitmix_draw(seed0, base + ((5 * i + 3) & (N - 1))) & mask;
    tv[i] = (u64)(i >> rep);
    cv[i] = gl_mul(r, r);
    dv[i] = gl_add(gl_mul(fa[i], r), fb[i]);
    atomicAdd((unsigned long long *)&mv[(size_t)r << rep], 1ULL);
}
// permutation AIR (kind 2): a = draws 0 .., b[i] = a[(5 i + 3) mod N], c = a^2
__global__ void __launch_bounds__(256) perm_kernel(u64 *__restrict__ tr, size_t N, u64 seed0) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const u64 a = splitmix_draw(seed0, i);
    tr[i] = a;
    tr[N + i] = splitmix_draw(seed0, (5 * i + 3) & (N - 1));
    tr[2 * N + i] = gl_mul(a, a);
}

int mix_width(int kind, int W) { return kind == 1 ? W : kind == 3 ? W - 8 : 0; }
bool shape_ok(int kind, int logn, int W) {
    if (logn < 1 || logn > 30) return false;
    switch (kind) {
        case 0: return W == 2;
        case 1: return W >= 3 && W <= 64 * MAX_CPL;
        case 2: return W == 3;
        case 3: return W >= 12 && W - 8 <= 64 * MAX_CPL;
        default: return false;
    }
}
int max_bind(int kind, int W) { return kind == 0 ? 2 : kind == 1 ? (W < 4 ? W : 4) : kind == 3 ? 6 : 0; }
int seg_log_of(int logn) { return logn < SEG_LOG ? logn : SEG_LOG; }

// first row of the wide mix of one chunk (draws 0 .. Ww-1, the first min(4, n_bind) dictated)
void mix_first_row(int Ww, u64 seed, const u64 *bind, int n_bind, u64 *row) {
    for (int i = 0; i < Ww; i++) row[i] = splitmix_draw(seed, (u64)i);
    for (int i = 0; i < n_bind && i < 4; i++) row[i] = bind[i];
}

int32_t run_checkpoints(zp_ctx *ctx, int logn, int Ww, int n, const u64 *d_init, u64 *d_ckpt) {
    const int sl = seg_log_of(logn), cpl = (Ww + 63) / 64;
    switch (cpl) {
        case 1: hipLaunchKernelGGL(mix_checkpoint_kernel<1>, dim3(n), dim3(64), 0, ctx->stream, d_init, d_ckpt, logn, sl, Ww); break;
        case 2: hipLaunchKernelGGL(mix_checkpoint_kernel<2>, dim3(n), dim3(64), 0, ctx->stream, d_init, d_ckpt, logn, sl, Ww); break;
        case 3: hipLaunchKernelGGL(mix_checkpoint_kernel<3>, dim3(n), dim3(64), 0, ctx->stream, d_init, d_ckpt, logn, sl, Ww); break;
        default: hipLaunchKernelGGL(mix_checkpoint_kernel<4>, dim3(n), dim3(64), 0, ctx->stream, d_init, d_ckpt, logn, sl, Ww); break;
    }
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
int32_t run_expand(zp_ctx *ctx, int logn, int Ww, const u64 *d_ckpt, u64 *d_trace) {
    const int sl = seg_log_of(logn), cpl = (Ww + 63) / 64;
    const unsigned nseg = (unsigned)(((size_t)1 << logn) >> sl);
    // measured (profiles/r5_synth_fill_ab.txt): 2^22 x 64 2.57 -> 1.63 ms (kind 3), 3.31 -> 1.30 ms (kind 1); at 2^20 rows the 256 one-wave blocks
    // do not fill the chip either way and the extra LDS pass loses 12-16 %: column runs from 2^21 rows
    if (ctx->tune_synth_rowwise == 1 || (ctx->tune_synth_rowwise == 0 && logn < 21)) {
        switch (cpl) {
            case 1: hipLaunchKernelGGL(mix_expand_kernel<1>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            case 2: hipLaunchKernelGGL(mix_expand_kernel<2>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            case 3: hipLaunchKernelGGL(mix_expand_kernel<3>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            default: hipLaunchKernelGGL(mix_expand_kernel<4>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
        }
    } else {
        switch (cpl) {
            case 1: hipLaunchKernelGGL(mix_expand_tiled_kernel<1>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            case 2: hipLaunchKernelGGL(mix_expand_tiled_kernel<2>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            case 3: hipLaunchKernelGGL(mix_expand_tiled_kernel<3>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
            default: hipLaunchKernelGGL(mix_expand_tiled_kernel<4>, dim3(nseg), dim3(64), 0, ctx->stream, d_ckpt, d_trace, logn, sl, Ww); break;
        }
    }
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

}  // namespace

extern "C" {

size_t zp_synth_checkpoint_words(int32_t kind, int32_t logn, int32_t W) {
    if (!shape_ok(kind, logn, W)) return 0;
    return (((size_t)1 << logn) >> seg_log_of(logn)) * (size_t)mix_width(kind, W);
}

int32_t zp_synth_checkpoints(zp_ctx *ctx, int32_t kind, int32_t logn, int32_t W, int32_t n_chunks, const uint64_t *h_seeds, const uint64_t *h_bind,
                             int32_t n_bind, uint64_t *d_ckpt) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "synth_checkpoints");
    ZP_ARG(ctx, shape_ok(kind, logn, W) && mix_width(kind, W) > 0, "kind / logn / W: no wide-mix columns of that shape");
    ZP_ARG(ctx, n_chunks >= 1 && n_chunks <= 4096 && h_seeds && d_ckpt, "n_chunks out of range / null pointer");
    ZP_ARG(ctx, n_bind >= 0 && n_bind <= max_bind(kind, W) && (n_bind == 0 || h_bind), "n_bind out of range");
    for (size_t i = 0; i < (size_t)n_chunks * n_bind; i++) ZP_ARG(ctx, h_bind[i] < GL_P, "bind value not canonical");
    const int Ww = mix_width(kind, W);
    std::vector<u64> rows((size_t)n_chunks * Ww);
    for (int c = 0; c < n_chunks; c++) mix_first_row(Ww, h_seeds[c], h_bind ? (const u64 *)h_bind + (size_t)c * n_bind : nullptr, n_bind, &rows[(size_t)c * Ww]);
    void *d_init = nullptr;
    ZP_TRY(zp_dev_alloc(ctx, rows.size() * sizeof(u64), &d_init));
    int32_t rc = zp_h2d(ctx, d_init, rows.data(), rows.size() * sizeof(u64));
    if (rc == ZP_OK) rc = run_checkpoints(ctx, logn, Ww, n_chunks, (const u64 *)d_init, (u64 *)d_ckpt);
    if (rc == ZP_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        ctx->err = "zp_synth_checkpoints: kernel failed";
        rc = ZP_ERR_HIP;
    }
    (void)zp_dev_free(ctx, d_init);
    return rc;
}

int32_t zp_synth_trace_device(zp_ctx *ctx, int32_t kind, int32_t logn, int32_t W, uint64_t seed, const uint64_t *h_bind, int32_t n_bind,
                              const uint64_t *d_ckpt, uint64_t *d_trace, uint64_t *h_pub) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "synth_trace_device");
    ZP_ARG(ctx, shape_ok(kind, logn, W) && d_trace && h_pub, "kind / logn / W out of range or null pointer");
    ZP_ARG(ctx, n_bind >= 0 && n_bind <= max_bind(kind, W) && (n_bind == 0 || h_bind), "n_bind out of range");
    for (int i = 0; i < n_bind; i++) ZP_ARG(ctx, h_bind[i] < GL_P, "bind value not canonical");
    const size_t N = (size_t)1 << logn;
    const unsigned g256 = (unsigned)((N + 255) / 256);
    u64 *tr = (u64 *)d_trace;
    const int Ww = mix_width(kind, W);
    if (kind == 2) {
        hipLaunchKernelGGL(perm_kernel, dim3(g256), dim3(256), 0, ctx->stream, tr, N, (u64)seed);
        ZP_HIP(ctx, hipGetLastError());
        h_pub[0] = splitmix_draw(seed, 0);
        return ZP_OK;
    }
    void *d_own = nullptr;                      // checkpoints made here when the caller brings none
    if (Ww > 0) {
        std::vector<u64> row(Ww);
        mix_first_row(Ww, seed, (const u64 *)h_bind, n_bind, row.data());
        for (int i = 0; i < (Ww < 4 ? Ww : 4); i++) h_pub[i] = row[i];
        const u64 *ck = (const u64 *)d_ckpt;
        if (!ck) {
            ZP_TRY(zp_dev_alloc(ctx, zp_synth_checkpoint_words(kind, logn, W) * sizeof(u64), &d_own));
            int32_t rc = zp_synth_checkpoints(ctx, kind, logn, W, 1, &seed, h_bind, n_bind, (uint64_t *)d_own);
            if (rc != ZP_OK) {
                (void)zp_dev_free(ctx, d_own);
                return rc;
            }
            ck = (const u64 *)d_own;
        }
        const int32_t rc = run_expand(ctx, logn, Ww, ck, tr);
        if (rc != ZP_OK) {
            if (d_own) (void)zp_dev_free(ctx, d_own);
            return rc;
        }
    }
    if (kind == 0 || kind == 3) {
        u64 a = splitmix_draw(seed, (u64)Ww), b = splitmix_draw(seed, (u64)Ww + 1);
        const int o = kind == 3 ? 4 : 0;       // bind slots of the Fibonacci pair
        if (n_bind > o) a = h_bind[o];
        if (n_bind > o + 1) b = h_bind[o + 1];
        u64 *fa = tr + (size_t)Ww * N, *fb = fa + N;
        hipLaunchKernelGGL(fib_kernel, dim3((unsigned)((N / FIB_BLOCK + 256) / 256)), dim3(256), 0, ctx->stream, fa, fb, N, a, b);
        u64 al, bl;
        fib_at(a, b, (u64)N - 1, al, bl);
        if (kind == 0) {
            h_pub[0] = a;
            h_pub[1] = b;
            h_pub[2] = bl;
        } else {
            h_pub[4] = a;
            h_pub[5] = b;
            h_pub[6] = bl;
            const int k = logn < 16 ? logn : 16, rep = logn - k;
            h_pub[7] = ((u64)1 << k) - 1;
            u64 *rv = fb + N, *qv = rv + N, *tv = qv + N, *mv = tv + N, *cv = mv + N, *dv = cv + N;
            if (hipMemsetAsync(mv, 0, N * sizeof(u64), ctx->stream) != hipSuccess) {
                if (d_own) (void)zp_dev_free(ctx, d_own);
                ctx->err = "zp_synth_trace_device: hipMemsetAsync failed";
                return ZP_ERR_HIP;
            }
            hipLaunchKernelGGL(range_kernel, dim3(g256), dim3(256), 0, ctx->stream, rv, qv, tv, mv, cv, dv, (const u64 *)fa, (const u64 *)fb, N, (u64)seed,
                               (u64)Ww + 2, k, rep);
        }
    }
    int32_t rc = ZP_OK;
    if (hipGetLastError() != hipSuccess) {
        ctx->err = "zp_synth_trace_device: launch failed";
        rc = ZP_ERR_HIP;
    }
    if (d_own) {                                // the expansion reads it: drain before it goes back
        if (hipStreamSynchronize(ctx->stream) != hipSuccess && rc == ZP_OK) {
            ctx->err = "zp_synth_trace_device: kernel failed";
            rc = ZP_ERR_HIP;
        }
        (void)zp_dev_free(ctx, d_own);
    }
    return rc;
}

}  // extern "C"
