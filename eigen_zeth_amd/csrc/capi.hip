// extern "C" surface of libzethprover.so -- see include/zeth_prover.h for the contract.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "ctx.hpp"
#include "poseidon_default_table.inc"

extern "C" {

const char *zp_version(void) { return "zethprover-mi355x 0.1 (gfx950)"; }

int32_t zp_device_count(void) {
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int32_t zp_create(zp_ctx **out, int32_t device) {
    if (!out) return ZP_ERR_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZP_ERR_HIP;
    if (device < 0 || device >= ndev) return ZP_ERR_ARG;
    if (hipSetDevice(device) != hipSuccess) return ZP_ERR_HIP;
    zp_ctx *ctx = new (std::nothrow) zp_ctx();
    if (!ctx) return ZP_ERR_NOMEM;
    ctx->device = device;
    memcpy(ctx->h_rc, ZP_POSEIDON_DEFAULT_RC, sizeof(ctx->h_rc));
    memcpy(ctx->h_mds, ZP_POSEIDON_DEFAULT_MDS, sizeof(ctx->h_mds));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
    // a ctx owns a non-blocking stream: two ctxs (e.g. prover + witness upload) overlap instead of serialising on
    // the legacy default stream
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return ZP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return ZP_OK;
}

void zp_destroy(zp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->plans) {
        if (kv.second.d_twl) (void)hipFree(kv.second.d_twl);
        if (kv.second.d_twh) (void)hipFree(kv.second.d_twh);
        if (kv.second.d_tws) (void)hipFree(kv.second.d_tws);
        if (kv.second.d_tw1) (void)hipFree(kv.second.d_tw1);
    }
    for (auto &c : ctx->cosets) {
        if (c.d_lo) (void)hipFree(c.d_lo);
        if (c.d_hi) (void)hipFree(c.d_hi);
    }
    for (int i = 0; i < 6; i++)
        if (ctx->scratch[i]) (void)hipFree(ctx->scratch[i]);
    for (auto &kv : ctx->prove_pool) (void)hipFree(kv.second);
    for (auto &kv : ctx->prove_fixed) (void)hipFree(kv.second);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    zpi_g16_cache_free(ctx);
    for (zp_ctx *&h : ctx->msm_helpers) {
        if (h) zp_destroy(h);
        h = nullptr;
    }
    if (ctx->msm_arena) (void)hipFree(ctx->msm_arena);
    if (ctx->d_rc) (void)hipFree(ctx->d_rc);
    if (ctx->d_mds) (void)hipFree(ctx->d_mds);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char *zp_last_error(zp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int32_t zp_set_stream(zp_ctx *ctx, void *hip_stream) {
    if (!ctx) return ZP_ERR_ARG;
    ctx->stream = (hipStream_t)hip_stream;
    return ZP_OK;
}

int32_t zp_get_stream(zp_ctx *ctx, void **hip_stream) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, hip_stream != nullptr, "null pointer");
    *hip_stream = (void *)ctx->stream;
    return ZP_OK;
}

int32_t zp_sync(zp_ctx *ctx) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

static void drop_plans(zp_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->plans) {
        if (kv.second.d_twl) (void)hipFree(kv.second.d_twl);
        if (kv.second.d_twh) (void)hipFree(kv.second.d_twh);
        if (kv.second.d_tws) (void)hipFree(kv.second.d_tws);
        if (kv.second.d_tw1) (void)hipFree(kv.second.d_tw1);
    }
    ctx->plans.clear();
}

int32_t zp_set_constants(zp_ctx *ctx, int32_t kind, const uint64_t *blob, size_t n) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, blob != nullptr, "null blob");
    switch (kind) {
        case ZP_CONST_ROOT32: {
            ZP_ARG(ctx, n == 1, "ROOT32 takes one element");
            u64 r = blob[0];
            ZP_ARG(ctx, r < GL_P, "root not canonical");
            // must have exact order 2^32:  r^(2^31) == -1
            u64 t = r;
            for (int i = 0; i < 31; i++) t = gl_mul(t, t);
            ZP_ARG(ctx, t == GL_P - 1, "root32 is not a primitive 2^32-th root of unity");
            if (r != ctx->root32) {
                ctx->root32 = r;
                drop_plans(ctx);
            }
            return ZP_OK;
        }
        case ZP_CONST_POSEIDON_RC:
            ZP_ARG(ctx, n == 360, "POSEIDON_RC takes 360 elements");
            for (size_t i = 0; i < n; i++) ZP_ARG(ctx, blob[i] < GL_P, "constant not canonical");
            memcpy(ctx->h_rc, blob, sizeof(ctx->h_rc));
            ctx->poseidon_dirty = true;
            return ZP_OK;
        case ZP_CONST_POSEIDON_MDS:
            ZP_ARG(ctx, n == 144, "POSEIDON_MDS takes 144 elements");
            for (size_t i = 0; i < n; i++) ZP_ARG(ctx, blob[i] < (1ULL << 28), "MDS entries must be < 2^28");
            memcpy(ctx->h_mds, blob, sizeof(ctx->h_mds));
            ctx->poseidon_dirty = true;
            return ZP_OK;
        case ZP_CONST_COSET_SHIFT:
            ZP_ARG(ctx, n == 1 && blob[0] != 0 && blob[0] < GL_P, "bad coset shift");
            ctx->coset_shift = blob[0];
            return ZP_OK;
        default:
            ctx->err = "unknown constant kind";
            return ZP_ERR_ARG;
    }
}

int32_t zp_get_constants(zp_ctx *ctx, int32_t kind, uint64_t *blob, size_t n) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, blob != nullptr, "null blob");
    switch (kind) {
        case ZP_CONST_ROOT32: ZP_ARG(ctx, n == 1, "size"); blob[0] = ctx->root32; return ZP_OK;
        case ZP_CONST_POSEIDON_RC: ZP_ARG(ctx, n == 360, "size"); memcpy(blob, ctx->h_rc, sizeof(ctx->h_rc)); return ZP_OK;
        case ZP_CONST_POSEIDON_MDS: ZP_ARG(ctx, n == 144, "size"); memcpy(blob, ctx->h_mds, sizeof(ctx->h_mds)); return ZP_OK;
        case ZP_CONST_COSET_SHIFT: ZP_ARG(ctx, n == 1, "size"); blob[0] = ctx->coset_shift; return ZP_OK;
        default: ctx->err = "unknown constant kind"; return ZP_ERR_ARG;
    }
}

// ---- memory
int32_t zp_dev_alloc(zp_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, d_ptr != nullptr, "null out pointer");
    *d_ptr = nullptr;
    if (bytes == 0) return ZP_OK;
    ZP_HIP(ctx, hipSetDevice(ctx->device));
    ZP_HIP(ctx, hipMalloc(d_ptr, bytes));
    return ZP_OK;
}
int32_t zp_dev_free(zp_ctx *ctx, void *d_ptr) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (!d_ptr) return ZP_OK;
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZP_HIP(ctx, hipFree(d_ptr));
    return ZP_OK;
}
int32_t zp_host_alloc(zp_ctx *ctx, size_t bytes, void **h_ptr) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, h_ptr != nullptr && bytes > 0, "null pointer / zero size");
    (void)hipSetDevice(ctx->device);
    if (hipHostMalloc(h_ptr, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        ctx->err = "zp_host_alloc: hipHostMalloc failed";
        return ZP_ERR_NOMEM;
    }
    return ZP_OK;
}
int32_t zp_host_free(zp_ctx *ctx, void *h_ptr) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (!h_ptr) return ZP_OK;
    ZP_HIP(ctx, hipHostFree(h_ptr));
    return ZP_OK;
}
int32_t zp_h2d(zp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (bytes == 0) return ZP_OK;
    ZP_ARG(ctx, d_dst && h_src, "null pointer");
    if (bytes <= ZP_SMALL_COPY) return zpi_h2d_small(ctx, d_dst, h_src, bytes);
    ZP_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}
int32_t zp_d2h(zp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (bytes == 0) return ZP_OK;
    ZP_ARG(ctx, h_dst && d_src, "null pointer");
    if (bytes <= ZP_SMALL_COPY) return zpi_d2h_small(ctx, h_dst, d_src, bytes);
    ZP_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}
int32_t zp_d2d(zp_ctx *ctx, void *d_dst, const void *d_src, size_t bytes) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (bytes == 0) return ZP_OK;
    ZP_ARG(ctx, d_dst && d_src, "null pointer");
    ZP_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZP_OK;
}

int32_t zp_dev_zero(zp_ctx *ctx, void *d_dst, size_t bytes) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    if (bytes == 0) return ZP_OK;
    ZP_ARG(ctx, d_dst != nullptr, "null pointer");
    ZP_HIP(ctx, hipMemsetAsync(d_dst, 0, bytes, ctx->stream));
    return ZP_OK;
}

// ---- N1
int32_t zp_ntt(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t W) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "ntt");
    NttRunOpts o;
    return zpi_ntt_run(ctx, (const u64 *)d_in, (u64 *)d_out, logn, W, false, o);
}
int32_t zp_intt(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t W) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "intt");
    NttRunOpts o;
    return zpi_ntt_run(ctx, (const u64 *)d_in, (u64 *)d_out, logn, W, true, o);
}

// ---- N2
int32_t zp_twiddle_rows(zp_ctx *ctx, uint64_t *d_rows, int32_t logn_row, int32_t W, uint64_t row0, int32_t logn_total,
                        int32_t inverse) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "twiddle_rows");
    ZP_ARG(ctx, logn_row >= 0 && logn_total >= logn_row && logn_total <= 32 && W >= 0, "sizes out of range");
    ZP_ARG(ctx, d_rows != nullptr || W == 0, "null pointer");
    ZP_ARG(ctx, row0 + (uint64_t)W <= (1ULL << (logn_total - logn_row)), "rows beyond N / 2^logn_row");
    if (W == 0) return ZP_OK;
    return zpi_twiddle_rows(ctx, (u64 *)d_rows, logn_row, W, row0, logn_total, inverse != 0);
}

int32_t zp_lde(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t *d_coef, int32_t logn,
               int32_t logb, int32_t W, uint64_t shift) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "lde");
    return zpi_lde(ctx, (const u64 *)d_in, (u64 *)d_out, (u64 *)d_coef, logn, logb, W, shift);
}

// ---- AIR plug-in support
int32_t zp_domain_tables(zp_ctx *ctx, int32_t logm, const uint64_t **d_lo, const uint64_t **d_hi, int32_t *lb) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, logm >= 0 && logm <= 32 && d_lo && d_hi && lb, "bad arguments");
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logm, false, &pl));
    *d_lo = (const uint64_t *)pl->d_twl;
    *d_hi = (const uint64_t *)pl->d_twh;
    *lb = pl->lb;
    return ZP_OK;
}

// host-only product mod p through the 128-bit multiply of the host CPU (the shared gl_mul is written for 32-bit GPU limbs)
static inline u64 host_mulmod(u64 a, u64 b) {
    const unsigned __int128 t = (unsigned __int128)a * b;
    return gl_reduce96((u64)t, (u32)(t >> 64), (u32)(t >> 96));
}

int32_t zp_synth_trace(int32_t kind, int32_t logn, int32_t W, uint64_t seed, uint64_t *h_trace, uint64_t *h_pub) {
    return zp_synth_trace_bound(kind, logn, W, seed, nullptr, 0, h_trace, h_pub);
}

// The same generators with the first n_bind starting values dictated by the caller instead of drawn from the seed: the
// AIRs constrain those cells to public inputs (L_first * (col - pub_i)), so a proof over such a witness names them --
// the service puts the limbs of the block statement there (state roots, transaction digest: service/statement.py), which
// binds every chunk proof to its block (prover.proto:80-91; provider.rs:315-330).
//   kind 0 (fib): bind[0..1] = a[0], b[0];  kind 1 (wide): bind[i] = c_i[0], i < min(4, W);
//   kind 3 (chunk): bind[0..3] = c_0..c_3[0], bind[4..5] = fa[0], fb[0].  Values must be canonical (< p).
int32_t zp_synth_trace_bound(int32_t kind, int32_t logn, int32_t W, uint64_t seed, const uint64_t *bind, int32_t n_bind,
                             uint64_t *h_trace, uint64_t *h_pub) {
    if (logn < 1 || logn > 30 || !h_trace || !h_pub) return ZP_ERR_ARG;
    if (n_bind < 0 || n_bind > 6 || (n_bind > 0 && !bind)) return ZP_ERR_ARG;
    for (int i = 0; i < n_bind; i++)
        if (bind[i] >= GL_P) return ZP_ERR_ARG;
    if (n_bind > (kind == 0 ? 2 : kind == 1 ? (W < 4 ? W : 4) : kind == 3 ? 6 : 0)) return ZP_ERR_ARG;
    const size_t N = (size_t)1 << logn;
    // splitmix64 -> canonical field elements
    auto next = [&seed]() {
        u64 z = (seed += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z ^= z >> 31;
        return z >= GL_P ? z - GL_P : z;
    };
    if (kind == 0) {
        if (W != 2) return ZP_ERR_ARG;
        u64 a = next(), b = next();
        if (n_bind > 0) a = bind[0];
        if (n_bind > 1) b = bind[1];
        h_pub[0] = a;
        h_pub[1] = b;
        for (size_t i = 0; i < N; i++) {
            h_trace[i] = a;
            h_trace[N + i] = b;
            const u64 t = gl_add(a, b);
            a = b;
            b = t;
        }
        h_pub[2] = h_trace[N + N - 1];
        return ZP_OK;
    }
    if (kind == 1) {
        if (W < 3) return ZP_ERR_ARG;
        std::vector<u64> cur(W), nx(W);
        for (int i = 0; i < W; i++) cur[i] = next();
        for (int i = 0; i < n_bind; i++) cur[i] = bind[i];
        for (int i = 0; i < (W < 4 ? W : 4); i++) h_pub[i] = cur[i];
        for (size_t r = 0; r < N; r++) {
            for (int i = 0; i < W; i++) h_trace[(size_t)i * N + r] = cur[i];
            for (int i = 0; i < W; i++)
                nx[i] = gl_add(gl_add(gl_mul(cur[i], cur[(i + 1) % W]), cur[(i + 2) % W]), (u64)i);
            cur.swap(nx);
        }
        return ZP_OK;
    }
    if (kind == 2) {  // permutation AIR: a random, b[i] = a[(5*i + 3) mod N], c = a^2
        if (W != 3) return ZP_ERR_ARG;
        for (size_t i = 0; i < N; i++) h_trace[i] = next();
        for (size_t i = 0; i < N; i++) {
            h_trace[N + i] = h_trace[(5 * i + 3) & (N - 1)];
            h_trace[2 * N + i] = gl_mul(h_trace[i], h_trace[i]);
        }
        h_pub[0] = h_trace[0];
        return ZP_OK;
    }
    if (kind == 3) {  // chunk AIR: wide mix | Fibonacci | range values, their permutation, table, multiplicities | products
        if (W < 12) return ZP_ERR_ARG;
        const int Ww = W - 8;
        std::vector<u64> cur(Ww), nx(Ww);
        for (int i = 0; i < Ww; i++) cur[i] = next();
        for (int i = 0; i < n_bind && i < 4; i++) cur[i] = bind[i];
        for (int i = 0; i < 4; i++) h_pub[i] = cur[i];
        // rows are produced 64 at a time into a small row-major block and written out column by column, so that the
        // column-major trace receives 512-byte runs instead of one 8-byte store per (row, column)
        {
            const size_t RB = N < 64 ? N : 64;
            std::vector<u64> blk((size_t)Ww * RB), ext(Ww + 2);
            for (size_t r0 = 0; r0 < N; r0 += RB) {
                for (size_t rr = 0; rr < RB; rr++) {
                    for (int i = 0; i < Ww; i++) {
                        blk[(size_t)i * RB + rr] = cur[i];
                        ext[i] = cur[i];
                    }
                    ext[Ww] = cur[0];            // wrap-around neighbours without a modulo in the inner loop
                    ext[Ww + 1] = cur[1 % Ww];
                    for (int i = 0; i < Ww; i++) cur[i] = gl_add(gl_add(host_mulmod(ext[i], ext[i + 1]), ext[i + 2]), (u64)i);
                }
                for (int i = 0; i < Ww; i++) memcpy(h_trace + (size_t)i * N + r0, &blk[(size_t)i * RB], RB * sizeof(u64));
            }
        }
        u64 *fa = (u64 *)h_trace + (size_t)Ww * N, *fb = fa + N, *rv = fb + N, *qv = rv + N, *tv = qv + N, *mv = tv + N,
            *cv = mv + N, *dv = cv + N;
        u64 a = next(), b = next();
        if (n_bind > 4) a = bind[4];
        if (n_bind > 5) b = bind[5];
        h_pub[4] = a;
        h_pub[5] = b;
        for (size_t i = 0; i < N; i++) {
            fa[i] = a;
            fb[i] = b;
            const u64 t = gl_add(a, b);
            a = b;
            b = t;
        }
        h_pub[6] = fb[N - 1];
        const int k = logn < 16 ? logn : 16;
        const int rep = logn - k;
        h_pub[7] = ((u64)1 << k) - 1;
        for (size_t i = 0; i < N; i++) {
            rv[i] = next() & (((u64)1 << k) - 1);
            tv[i] = (u64)(i >> rep);
            mv[i] = 0;
        }
        for (size_t i = 0; i < N; i++) {
            mv[(size_t)rv[i] << rep] += 1;
            qv[i] = rv[(5 * i + 3) & (N - 1)];
            cv[i] = gl_mul(rv[i], rv[i]);
            dv[i] = gl_add(gl_mul(fa[i], rv[i]), fb[i]);
        }
        return ZP_OK;
    }
    return ZP_ERR_ARG;
}

// ---- host conveniences
int32_t zp_ntt_host(zp_ctx *ctx, uint64_t *h_cols, int32_t logn, int32_t W, int32_t inverse) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, h_cols != nullptr || W == 0, "null host pointer");
    ZP_ARG(ctx, logn >= 0 && logn <= 32 && W >= 0, "logn/W out of range");
    if (W == 0) return ZP_OK;
    const size_t bytes = ((size_t)W << logn) * sizeof(u64);
    void *d = nullptr;
    ZP_TRY(zp_dev_alloc(ctx, bytes, &d));
    int32_t rc = zp_h2d(ctx, d, h_cols, bytes);
    if (rc == ZP_OK) {
        NttRunOpts o;
        rc = zpi_ntt_run(ctx, (u64 *)d, (u64 *)d, logn, W, inverse != 0, o);
    }
    if (rc == ZP_OK) rc = zp_d2h(ctx, h_cols, d, bytes);
    (void)zp_dev_free(ctx, d);
    return rc;
}

int32_t zp_lde_host(zp_ctx *ctx, const uint64_t *h_in, uint64_t *h_out, int32_t logn, int32_t logb,
                    int32_t W, uint64_t shift) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, (h_in && h_out) || W == 0, "null host pointer");
    ZP_ARG(ctx, logn >= 0 && logb >= 0 && logn + logb <= 32 && W >= 0, "logn/logb/W out of range");
    if (W == 0) return ZP_OK;
    const size_t bin = ((size_t)W << logn) * sizeof(u64), bout = bin << logb;
    void *di = nullptr, *dout = nullptr;
    ZP_TRY(zp_dev_alloc(ctx, bin, &di));
    int32_t rc = zp_dev_alloc(ctx, bout, &dout);
    if (rc == ZP_OK) rc = zp_h2d(ctx, di, h_in, bin);
    if (rc == ZP_OK) rc = zp_lde(ctx, (const uint64_t *)di, (uint64_t *)dout, nullptr, logn, logb, W, shift);
    if (rc == ZP_OK) rc = zp_d2h(ctx, h_out, dout, bout);
    (void)zp_dev_free(ctx, di);
    (void)zp_dev_free(ctx, dout);
    return rc;
}

int32_t zp_set_tuning(zp_ctx *ctx, const char *key, int32_t value) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, key != nullptr, "null key");
    if (!strcmp(key, "ntt_logt")) ctx->tune_logt = value;
    else if (!strcmp(key, "ntt_tpw")) ctx->tune_tpw = value;
    else if (!strcmp(key, "ntt_logt9")) ctx->tune_logt9 = value;
    else if (!strcmp(key, "ntt_logt12")) ctx->tune_logt12 = value;
    else if (!strcmp(key, "msm_chunk_log")) ctx->tune_msm_chunk_log = value;
    else if (!strcmp(key, "msm_c")) ctx->tune_msm_c = value;
    else if (!strcmp(key, "ntt_chunk_log")) ctx->tune_ntt_chunk_log = value;
    else if (!strcmp(key, "ntt_maxl")) ctx->tune_ntt_maxl = value;
    else if (!strcmp(key, "ntt_order")) ctx->tune_ntt_order = value;
    else if (!strcmp(key, "ntt_tw1")) ctx->tune_ntt_tw1 = value;
    else if (!strcmp(key, "ntt_limb")) ctx->tune_ntt_limb = value;
    else if (!strcmp(key, "lde_seam")) ctx->tune_lde_seam = value;
    else if (!strcmp(key, "g16_parallel")) ctx->tune_g16_parallel = value;
    else if (!strcmp(key, "seam_tpw")) ctx->tune_seam_tpw = value;
    else if (!strcmp(key, "lde_seam_plans")) ctx->tune_lde_seam_plans = value;
    else if (!strcmp(key, "merkle_top_wave")) ctx->tune_merkle_top_wave = value;
    else if (!strcmp(key, "ntt_small_wave")) ctx->tune_ntt_small_wave = value;
    else if (!strcmp(key, "synth_rowwise")) ctx->tune_synth_rowwise = value;
    else if (!strcmp(key, "p254_scaled")) ctx->tune_p254_scaled = value;
    else if (!strcmp(key, "p254_block")) ctx->tune_p254_block = value;
    else if (!strcmp(key, "fri_fold_lanes")) ctx->tune_fri_fold_lanes = value;
    else if (!strcmp(key, "copy_grid")) ctx->tune_copy_grid = value;
    else if (!strcmp(key, "copy_nt")) ctx->tune_copy_nt = value;
    else if (!strcmp(key, "copy_block")) ctx->tune_copy_block = value;
    else if (!strcmp(key, "copy_unroll")) ctx->tune_copy_unroll = value;
    else if (!strcmp(key, "merkle_coop_log")) ctx->tune_merkle_coop_log = value;
    else if (!strcmp(key, "p254_bulk_log")) ctx->tune_p254_bulk_log = value;
    else { ctx->err = "unknown tuning key"; return ZP_ERR_ARG; }
    return ZP_OK;
}

int32_t zp_set_profiling(zp_ctx *ctx, int32_t on) {
    if (!ctx) return ZP_ERR_ARG;
    ctx->profiling = on != 0;
    return ZP_OK;
}

int32_t zp_get_pass_timings(zp_ctx *ctx, float *ms, int32_t *radix_log, int32_t cap, int32_t *count) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, count != nullptr && cap >= 0 && (cap == 0 || (ms && radix_log)), "bad arguments");
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (auto &ev : ctx->pass_events) {
        if (n < cap) {
            float t = 0.f;
            ZP_HIP(ctx, hipEventElapsedTime(&t, ev.a, ev.b));
            ms[n] = t;
            radix_log[n] = ev.radix_log;
            n++;
        }
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    ctx->pass_events.clear();
    *count = n;
    return ZP_OK;
}

int32_t zp_stage_timings(zp_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, buf && buflen > 2, "null buffer");
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    std::string s = "[";
    bool first = true;
    for (auto &ev : ctx->stage_events) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, ev.a, ev.b) == hipSuccess) {
            char tmp[128];
            snprintf(tmp, sizeof(tmp), "%s{\"stage\": \"%s\", \"ms\": %.4f}", first ? "" : ", ", ev.name, t);
            s += tmp;
            first = false;
        }
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    ctx->stage_events.clear();
    s += "]";
    ZP_ARG(ctx, s.size() + 1 <= buflen, "buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return ZP_OK;
}

int32_t zp_ntt_plan_json(zp_ctx *ctx, int32_t logn, char *buf, size_t buflen) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, buf && buflen > 0, "null buffer");
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logn, false, &pl));
    std::string s = "{\"logn\": " + std::to_string(logn) + ", \"passes\": [";
    for (int i = 0; i < pl->npass; i++) {
        const NttPass &p = pl->pass[i];
        if (i) s += ", ";
        s += "{\"radix_log\": " + std::to_string(p.L) + ", \"rounds\": [" + std::to_string(p.A1) + ", " +
             std::to_string(p.A2) + ", " + std::to_string(p.A3) + "], \"tile\": " + std::to_string(p.L == 8 ? (1 << ctx->tune_logt) : p.L == 9 ? (1 << ctx->tune_logt9) : (1 << p.logT)) + "}";
    }
    s += "], \"first_pass_table\": ";   // the transposing pass multiplies by the full precomputed table (MODE 3) instead of per-lane chains
    s += (pl->npass >= 1 && logn > 12 && !pl->tw1_unavailable && logn <= ctx->tune_ntt_tw1 && logn <= 28 && pl->pass[0].A3 == 0 && pl->pass[0].L >= 7) ? "true" : "false";
    s += ", \"small_kernel\": ";
    s += (logn <= 12) ? "true" : "false";
    if (logn + 1 <= 32) {     // the extension (blow-up 2) of columns of this size, as the provers issue it (no coefficient store since round 5)
        std::string l;
        ZP_TRY(zpi_lde_plan_json(ctx, logn, 0, &l));
        s += ", \"lde\": " + l;
    }
    s += "}";
    ZP_ARG(ctx, s.size() + 1 <= buflen, "buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return ZP_OK;
}

int32_t zp_device_info_json(zp_ctx *ctx, char *buf, size_t buflen) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, buf && buflen > 0, "null buffer");
    hipDeviceProp_t prop;
    ZP_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    size_t fr = 0, tot = 0;
    ZP_HIP(ctx, hipMemGetInfo(&fr, &tot));
    char tmp[512];
    snprintf(tmp, sizeof(tmp),
             "{\"name\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"total_mem\": %zu, \"free_mem\": %zu}",
             prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate, tot, fr);
    ZP_ARG(ctx, strlen(tmp) + 1 <= buflen, "buffer too small");
    memcpy(buf, tmp, strlen(tmp) + 1);
    return ZP_OK;
}

}  // extern "C"
