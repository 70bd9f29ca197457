// BN128-hash mode ("final STARK before the Groth16 wrap"): Poseidon over the BN254 scalar field,
// x^5 S-box, 8 full + R_P partial rounds (t = 3: 57, t = 17: 68), and the 16-ary Merkle tree over Goldilocks columns
// packed three to a field element.  The request this serves is GenFinalProof (proto/prover/v1/prover.proto:130-148,
// src/prover/provider.rs:472-503); the reference holds none of it.
// Anchor: with the Grain-LFSR tables of eigen_zeth_amd/poseidon_constants.py the t = 3 instance reproduces the published
// value poseidon([1, 2]) = 0x115cc0f5...189a (tests/test_poseidon_constants.py, tests/test_gpu_bn254_hash.py).
//
// Mapping: T lanes per permutation (lane e owns state element e in nine 29-bit Montgomery limbs), floor(64/T)
// permutations per 64-lane workgroup; a round is  ARK -> S-box (lane-local) -> state to LDS -> each lane one row of the
// dense t x t matrix.  Textbook schedule (no sparse partial-round matrices yet): 22 k field products per t = 17
// permutation, VALU-bound by construction.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.hpp"
#include "fr254.hpp"

namespace {

struct P254Table {    // device tables of one instance, Montgomery form, 9 x u32 per element
    int t = 0, rp = 0;
    u32 *d_rc = nullptr;    // [(8 + rp) * t][9]
    u32 *d_mds = nullptr;   // [t * t][9]   row-major: out[i] = sum_j mds[i][j] * in[j]
};
P254Table g_tables[2];      // [0]: t = 3, [1]: t = 17  (per process; the tables are public constants)

__device__ __forceinline__ fr fr_load(const u32 *p) {
    fr r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}
__device__ __forceinline__ fr sbox5(const fr &x) {
    const fr x2 = fr_mul(x, x), x4 = fr_mul(x2, x2);
    return fr_mul(x4, x);
}

// one permutation of the PPW states a 64-lane workgroup holds; lane = (slot q, element e); sh[q][j] = element j of slot q
template <int T>
__device__ __forceinline__ fr perm_coop(fr s, int q, int e, bool on, u32 (*sh)[T][9], int rp, const u32 *rc, const u32 *mds) {
    const int rounds = 8 + rp;
    for (int r = 0; r < rounds; r++) {
        if (on) {
            s = fr_add(s, fr_load(rc + ((size_t)r * T + e) * 9));
            if (r < 4 || r >= 4 + rp || e == 0) s = sbox5(s);
#pragma unroll
            for (int i = 0; i < 9; i++) sh[q][e][i] = s.l[i];
        }
        __syncthreads();
        if (on) {
            fr acc = fr_zero();
            for (int j = 0; j < T; j++) {
                fr v;
#pragma unroll
                for (int i = 0; i < 9; i++) v.l[i] = sh[q][j][i];
                acc = fr_add(acc, fr_mul(fr_load(mds + ((size_t)e * T + j) * 9), v));
            }
            s = acc;
        }
        __syncthreads();
    }
    return s;
}

// states: u64[count][T][4] standard form, permuted in place
template <int T>
__global__ void __launch_bounds__(64) poseidon254_perm_kernel(u64 *states, size_t count, int rp, const u32 *rc, const u32 *mds) {
    constexpr int PPW = 64 / T;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t idx = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && idx < count;
    fr s = fr_zero();
    if (on) {
        u64 w[4];
#pragma unroll
        for (int k = 0; k < 4; k++) w[k] = states[(idx * T + e) * 4 + k];
        s = fr_to_mont(fr_from_u64(w));
    }
    s = perm_coop<T>(s, q, e, on, sh, rp, rc, mds);
    if (on) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) states[(idx * T + e) * 4 + k] = w[k];
    }
}

// 16-ary Merkle tree, t = 17: state = [capacity, 16 inputs]; digest = state[0] after the permutation.
// leaves: row i of the Goldilocks matrix cols[W][M], three values per field element (a + b 2^64 + c 2^128), sponge over
// blocks of 16 elements with the digest as the next capacity.  nodes: 16 child digests (missing children = 0).
__global__ void __launch_bounds__(64) merkle16_leaves_kernel(const u64 *__restrict__ cols, size_t M, int W, u64 *__restrict__ tree, int rp,
                                                            const u32 *rc, const u32 *mds) {
    constexpr int T = 17, PPW = 3;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t i = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && i < M;
    const int ne = (W + 2) / 3;                        // packed elements per row
    fr s = fr_zero();                                  // capacity 0
    for (int off = 0; off < ne || off == 0; off += 16) {
        if (on && e >= 1) {
            const int k = off + e - 1;                 // packed element index
            u64 w[4] = {0, 0, 0, 0};
            if (k < ne) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    if (3 * k + c < W) w[c] = cols[(size_t)(3 * k + c) * M + i];
            }
            s = fr_to_mont(fr_from_u64(w));
        }
        s = perm_coop<T>(s, q, e, on, sh, rp, rc, mds);
        // the digest (element 0) is the capacity of the next block: lane e = 0 already holds it
    }
    if (on && e == 0) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) tree[i * 4 + k] = w[k];
    }
}

__global__ void __launch_bounds__(64) merkle16_level_kernel(const u64 *__restrict__ prev, size_t nprev, u64 *__restrict__ next, size_t nnext, int rp,
                                                           const u32 *rc, const u32 *mds) {
    constexpr int T = 17, PPW = 3;
    __shared__ u32 sh[PPW][T][9];
    const int q = threadIdx.x / T, e = threadIdx.x % T;
    const size_t i = (size_t)blockIdx.x * PPW + q;
    const bool on = q < PPW && i < nnext;
    fr s = fr_zero();
    if (on && e >= 1) {
        const size_t c = i * 16 + (e - 1);
        u64 w[4] = {0, 0, 0, 0};
        if (c < nprev) {
#pragma unroll
            for (int k = 0; k < 4; k++) w[k] = prev[c * 4 + k];
        }
        s = fr_to_mont(fr_from_u64(w));
    }
    s = perm_coop<T>(s, q, e, on, sh, rp, rc, mds);
    if (on && e == 0) {
        u64 w[4];
        fr_to_u64(fr_from_mont(s), w);
#pragma unroll
        for (int k = 0; k < 4; k++) next[i * 4 + k] = w[k];
    }
}

int32_t table_for(zp_ctx *ctx, int t, P254Table **out) {
    P254Table *tb = t == 3 ? &g_tables[0] : t == 17 ? &g_tables[1] : nullptr;
    ZP_ARG(ctx, tb != nullptr, "Poseidon-BN254 width must be 3 or 17");
    ZP_ARG(ctx, tb->d_rc != nullptr, "Poseidon-BN254 tables not installed (zp_set_poseidon_bn254)");
    *out = tb;
    return ZP_OK;
}

}  // namespace

extern "C" {

int32_t zp_set_poseidon_bn254(zp_ctx *ctx, int32_t t, int32_t rp, const uint64_t *h_rc, const uint64_t *h_mds) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, (t == 3 || t == 17) && rp >= 1 && rp <= 128 && h_rc && h_mds, "bad arguments (t must be 3 or 17)");
    const size_t nrc = (size_t)(8 + rp) * t, nm = (size_t)t * t;
    std::vector<u32> rc(nrc * 9), md(nm * 9);
    for (size_t i = 0; i < nrc + nm; i++) {
        const uint64_t *src = i < nrc ? h_rc + 4 * i : h_mds + 4 * (i - nrc);
        u64 w[4] = {src[0], src[1], src[2], src[3]};
        ZP_ARG(ctx, fr_is_canonical_u64(w), "constant not reduced mod r");
        const fr m = fr_to_mont(fr_from_u64(w));
        memcpy((i < nrc ? rc.data() + 9 * i : md.data() + 9 * (i - nrc)), m.l, 36);
    }
    P254Table *tb = t == 3 ? &g_tables[0] : &g_tables[1];
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (tb->d_rc) { (void)hipFree(tb->d_rc); (void)hipFree(tb->d_mds); tb->d_rc = tb->d_mds = nullptr; }
    ZP_HIP(ctx, hipMalloc((void **)&tb->d_rc, rc.size() * 4));
    ZP_HIP(ctx, hipMalloc((void **)&tb->d_mds, md.size() * 4));
    ZP_HIP(ctx, hipMemcpy(tb->d_rc, rc.data(), rc.size() * 4, hipMemcpyHostToDevice));
    ZP_HIP(ctx, hipMemcpy(tb->d_mds, md.data(), md.size() * 4, hipMemcpyHostToDevice));
    tb->t = t;
    tb->rp = rp;
    return ZP_OK;
}

int32_t zp_poseidon_bn254_perm(zp_ctx *ctx, uint64_t *d_states, size_t count, int32_t t) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poseidon_bn254_perm");
    P254Table *tb;
    ZP_TRY(table_for(ctx, t, &tb));
    if (count == 0) return ZP_OK;
    ZP_ARG(ctx, d_states != nullptr, "null device pointer");
    if (t == 3)
        hipLaunchKernelGGL(poseidon254_perm_kernel<3>, dim3((unsigned)((count + 20) / 21)), dim3(64), 0, ctx->stream, (u64 *)d_states, count,
                           tb->rp, tb->d_rc, tb->d_mds);
    else
        hipLaunchKernelGGL(poseidon254_perm_kernel<17>, dim3((unsigned)((count + 2) / 3)), dim3(64), 0, ctx->stream, (u64 *)d_states, count,
                           tb->rp, tb->d_rc, tb->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

// d_tree: u64[nodes][4], leaves first (M of them), then ceil(M/16), ... down to the single root (the last 4 words)
size_t zp_merkle16_nodes(size_t M) {
    size_t n = M, total = M;
    while (n > 1) { n = (n + 15) / 16; total += n; }
    return total;
}

int32_t zp_merkle16_commit_bn254(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, uint64_t *d_tree) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "merkle16_commit_bn254");
    P254Table *tb;
    ZP_TRY(table_for(ctx, 17, &tb));
    ZP_ARG(ctx, d_cols && d_tree && M >= 1 && W >= 1, "bad arguments");
    hipLaunchKernelGGL(merkle16_leaves_kernel, dim3((unsigned)((M + 2) / 3)), dim3(64), 0, ctx->stream, (const u64 *)d_cols, M, (int)W,
                       (u64 *)d_tree, tb->rp, tb->d_rc, tb->d_mds);
    ZP_HIP(ctx, hipGetLastError());
    size_t n = M, off = 0;
    while (n > 1) {
        const size_t nn = (n + 15) / 16;
        hipLaunchKernelGGL(merkle16_level_kernel, dim3((unsigned)((nn + 2) / 3)), dim3(64), 0, ctx->stream, (const u64 *)d_tree + off * 4, n,
                           (u64 *)d_tree + (off + n) * 4, nn, tb->rp, tb->d_rc, tb->d_mds);
        ZP_HIP(ctx, hipGetLastError());
        off += n;
        n = nn;
    }
    return ZP_OK;
}

// opening of leaf idx: for every level the 16 digests of the group the path passes through (the verifier re-hashes the
// group with its own digest in place): h_path u64[levels][16][4], bottom-up
int32_t zp_merkle16_open_bn254(zp_ctx *ctx, const uint64_t *d_tree, size_t M, size_t idx, uint64_t *h_path) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, d_tree && h_path && idx < M, "bad arguments");
    size_t n = M, off = 0, pos = idx;
    int lvl = 0;
    while (n > 1) {
        const size_t g0 = (pos / 16) * 16;
        uint64_t grp[64];
        memset(grp, 0, sizeof(grp));
        const size_t have = g0 + 16 <= n ? 16 : n - g0;
        ZP_TRY(zpi_d2h_small(ctx, grp, d_tree + (off + g0) * 4, have * 32));
        memcpy(h_path + (size_t)lvl * 64, grp, sizeof(grp));
        off += n;
        n = (n + 15) / 16;
        pos /= 16;
        lvl++;
    }
    return ZP_OK;
}

}  // extern "C"
