// STARK stages around the committed columns (SURVEY.md 8a N4/N5 support): out-of-domain
// evaluation, DEEP quotient, row gathers and batched Merkle openings.  No reference counterpart in
// /root/reference (the prover behind src/prover/provider.rs:358-377 is external); definitions follow
// the public DEEP-FRI construction, checked against oracle/ and an independent verifier in tests/.
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "ctx.hpp"

namespace {

// ---- p_c(z) for W polynomials with base-field coefficients at an F_{p^3} point --------------------
// a block owns EV_CH = 256*EV_J consecutive coefficients; lane t reads c[256 j + t] (coalesced rows of 2 KiB),
// accumulates sum_j c[256 j + t] z^(256 j) unreduced (z^(256 j) wave-uniform), multiplies by z^t from a
// 256-entry table and the block reduces in LDS; one partial per (chunk, column).
#define EV_J 32
#define EV_CH (256 * EV_J)
struct EvalArgs {
    const u64 *coef;
    u64 n;
    const u64 *zlow;   // z^(256 j), j < EV_J   [EV_J][3]
    const u64 *zmid;   // z^t, t < 256          [256][3]
    u64 *partial;      // [chunks][W][3]
    int W;
};

__global__ void __launch_bounds__(256) poly_eval_ext_kernel(EvalArgs a) {
    __shared__ u64 red[3][256];
    const int t = threadIdx.x;
    const u64 chunk = blockIdx.x, col = blockIdx.y;
    const u64 base = chunk * EV_CH + (u64)t;
    const u64 *c = a.coef + col * a.n;
    gl_acc s0 = gl_acc_zero(), s1 = gl_acc_zero(), s2 = gl_acc_zero();   // unreduced: one reduction per EV_J terms
#pragma unroll 8
    for (int j = 0; j < EV_J; j++) {
        const u64 i = base + (u64)j * 256;
        const u64 v = (i < a.n) ? c[i] : 0ULL;
        gl_acc_mac(s0, v, a.zlow[j * 3 + 0]);
        gl_acc_mac(s1, v, a.zlow[j * 3 + 1]);
        gl_acc_mac(s2, v, a.zlow[j * 3 + 2]);
    }
    e3 r = e3_mul(e3_make(gl_acc_reduce(s0), gl_acc_reduce(s1), gl_acc_reduce(s2)), e3_make(a.zmid[t * 3], a.zmid[t * 3 + 1], a.zmid[t * 3 + 2]));
    red[0][t] = r.c[0]; red[1][t] = r.c[1]; red[2][t] = r.c[2];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) {
#pragma unroll
            for (int k = 0; k < 3; k++) red[k][t] = gl_add(red[k][t], red[k][t + s]);
        }
        __syncthreads();
    }
    if (t < 3) a.partial[(chunk * a.W + col) * 3 + t] = red[t][0];
}

// ---- p_c(z) (and p_c(z w)) FROM VALUES: barycentric evaluation ------------------------------------------
// Column c holds P_c(y) = p_c(shift y), deg P_c < n = 2^logn, at y_i = w^i (w of order n), row i at cols[c * cs + i * rs]:
//     P(y) = (y^n - 1) / n * sum_i P(w^i) u_i(y),       u_i(y) = w^i / (y - w^i)
// and, the domain being a group,  sum_i P(w^i) w^i / (y w - w^i) = sum_j P(w^(j+1)) u_j(y): the evaluation at the NEXT point y w uses
// the same weights on the column rotated by one row.  So one pass over the resident extension gives both out-of-domain evaluations of
// a STARK -- no coefficient buffer is kept for them (round 4 kept 8 N bytes per column and read them twice, csrc/prove.hip), and the
// LDE that made the columns may run without its coefficient store (the fused seam kernel, csrc/ntt.hip).
// (1) bary_weights_kernel: u[3][n] for one y, two rows per lane and ONE base-field inversion for both (as deep_quotient_kernel);
//     a vanishing denominator (y on the domain: excluded by the protocol) raises a flag the host turns into an error.
// (2) bary_dot_kernel: a block owns BY_ROWS rows x BY_CB columns: a lane loads u_i once and uses it for the BY_CB columns (each
//     twice: row i for y, row i + 1 for y w); unreduced accumulators, one reduction per lane; lanes are summed with wave
//     shuffles, the four waves through LDS; one partial per (block, column).  (3) bary_reduce_kernel sums the partials of a column.
//     The accumulators are gl_acc's device form (gl.hpp, round 5: three sums of 32 x 32-bit partial products with carry counts, 8 instructions
//     per multiply-accumulate; with the 160-bit form's 22 the kernel was bound by them: 130 instructions per value, 1.2 TB/s on the columns'
//     bytes); the sums meet once per lane, after BY_J rows.
#define BY_CB 4
#define BY_J 64
#define BY_ROWS (256 * BY_J)
struct BaryWArgs {
    u64 *u;                 // [3][n]
    unsigned int *flag;
    const u64 *twl, *twh;   // w^e
    u64 y[3];
    u64 n;
    int lb;
};
__global__ void __launch_bounds__(256) bary_weights_kernel(BaryWArgs a) {
    const u64 i0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i0 >= a.n) return;
    const bool two = i0 + 1 < a.n;
    const u64 lm = (1ULL << a.lb) - 1;
    const u64 w0 = gl_mul(a.twl[i0 & lm], a.twh[i0 >> a.lb]);
    const u64 w1 = two ? gl_mul(a.twl[(i0 + 1) & lm], a.twh[(i0 + 1) >> a.lb]) : w0;
    const e3 d0 = e3_make(gl_sub(a.y[0], w0), a.y[1], a.y[2]), d1 = e3_make(gl_sub(a.y[0], w1), a.y[1], a.y[2]);
    u64 det0, det1;
    const e3 adj0 = e3_adj(d0, &det0), adj1 = e3_adj(d1, &det1);
    if (det0 == 0 || det1 == 0) atomicOr(a.flag, 1u);
    const u64 m0 = det0 ? det0 : 1, m1 = det1 ? det1 : 1;
    const u64 inv01 = gl_inv(gl_mul(m0, m1));
    const e3 u0 = e3_scale(adj0, gl_mul(w0, gl_mul(inv01, m1))), u1 = e3_scale(adj1, gl_mul(w1, gl_mul(inv01, m0)));
#pragma unroll
    for (int c = 0; c < 3; c++) {
        a.u[(u64)c * a.n + i0] = u0.c[c];
        if (two) a.u[(u64)c * a.n + i0 + 1] = u1.c[c];
    }
}

struct BaryArgs {
    const u64 *cols;
    const u64 *u;       // [3][n]
    u64 *partial;       // [blocks][W][6]
    u64 cs, rs, n, blocks;
    int W, groups;
};
__device__ __forceinline__ u64 wave_sum_gl(u64 v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = gl_add(v, (u64)__shfl_xor((unsigned long long)v, o));
    return v;
}
template <bool NEXT>
__global__ void __launch_bounds__(256) bary_dot_kernel(BaryArgs a) {
    __shared__ u64 red[4][BY_CB * 6];
    const int t = threadIdx.x;
    // one-dimensional grid, XCD-aware: workgroup id -> (XCD x = id mod 8, j = id / 8); the column groups of ONE row chunk run next to each
    // other on ONE XCD (group = j mod groups, chunk = (j / groups) 8 + x), so the chunk's 96 KiB of weights come from HBM once and are L2 hits
    // for the other groups (a two-dimensional grid streamed the whole weight vector once per column group)
    const u64 xcd = blockIdx.x & 7u, jj = blockIdx.x >> 3;
    const u64 chunk = (jj / (u64)a.groups) * 8 + xcd;
    if (chunk >= a.blocks) return;
    const int c0 = (int)(jj % (u64)a.groups) * BY_CB;
    const u64 row0 = chunk * BY_ROWS;
    gl_acc acc[BY_CB][NEXT ? 6 : 3];
#pragma unroll
    for (int c = 0; c < BY_CB; c++)
#pragma unroll
        for (int k = 0; k < (NEXT ? 6 : 3); k++) acc[c][k] = gl_acc_zero();
    // The sum for the NEXT point, sum_j P(w^(j+1)) u_j, is taken as sum_i P(w^i) u_(i-1): the WEIGHTS are read at two rows (the second one is
    // the neighbouring lane's first: a cache hit), every column value once.
    // (a column group's last column may lie beyond W: it reads column W - 1 again -- no branch in the loop, so that every load of a row is
    // in flight before its first product -- and its sums are dropped below)
    const u64 *colp[BY_CB];
#pragma unroll
    for (int c = 0; c < BY_CB; c++) colp[c] = a.cols + (u64)(c0 + c < a.W ? c0 + c : a.W - 1) * a.cs;
#pragma unroll 2
    for (int j = 0; j < BY_J; j++) {
        const u64 i = row0 + (u64)j * 256 + t;
        if (i < a.n) {
            const u64 u0 = a.u[i], u1 = a.u[a.n + i], u2 = a.u[2 * a.n + i];
            u64 p0 = 0, p1 = 0, p2 = 0;
            if constexpr (NEXT) {
                const u64 ip = (i + a.n - 1) & (a.n - 1);
                p0 = a.u[ip]; p1 = a.u[a.n + ip]; p2 = a.u[2 * a.n + ip];
            }
            u64 v[BY_CB];
#pragma unroll
            for (int c = 0; c < BY_CB; c++) v[c] = colp[c][i * a.rs];
#pragma unroll
            for (int c = 0; c < BY_CB; c++) {
                gl_acc_mac(acc[c][0], v[c], u0);
                gl_acc_mac(acc[c][1], v[c], u1);
                gl_acc_mac(acc[c][2], v[c], u2);
                if constexpr (NEXT) {
                    gl_acc_mac(acc[c][3], v[c], p0);
                    gl_acc_mac(acc[c][4], v[c], p1);
                    gl_acc_mac(acc[c][5], v[c], p2);
                }
            }
        }
    }
    const int wv = t >> 6, ln = t & 63;
#pragma unroll
    for (int c = 0; c < BY_CB; c++)
#pragma unroll
        for (int k = 0; k < (NEXT ? 6 : 3); k++) {
            const u64 r = wave_sum_gl(gl_acc_reduce(acc[c][k]));
            if (ln == 0) red[wv][c * 6 + k] = r;
        }
    __syncthreads();
    if (t < BY_CB * 6) {
        const int c = t / 6, k = t % 6;
        if (c0 + c < a.W && (NEXT || k < 3)) {
            const u64 r = gl_add(gl_add(red[0][t], red[1][t]), gl_add(red[2][t], red[3][t]));
            a.partial[(chunk * a.W + (c0 + c)) * 6 + k] = r;
        }
    }
}
// out[w * 6 + k] = sum over blocks of partial[b][w][k]; one workgroup per (column, component)
__global__ void __launch_bounds__(256) bary_reduce_kernel(const u64 *partial, u64 *out, u64 blocks, int W6) {
    __shared__ u64 red[4];
    const int t = threadIdx.x;
    u64 s = 0;
    for (u64 b = t; b < blocks; b += 256) s = gl_add(s, partial[b * (u64)W6 + blockIdx.x]);
    s = wave_sum_gl(s);
    if ((t & 63) == 0) red[t >> 6] = s;
    __syncthreads();
    if (t == 0) out[blockIdx.x] = gl_add(gl_add(red[0], red[1]), gl_add(red[2], red[3]));
}

// ---- DEEP quotient ---------------------------------------------------------------------------------
// F(x) = sum_{k<Wa+Wb} g^k (p_k(x) - e_k)/(x - z) + sum_{k<nnext} g^(Wa+Wb+k) (p_k(x) - e'_k)/(x - zw)
// on x = shift * w_M^r.  lane = row; column reads are coalesced; g^k and the constant terms are uniform.
struct DeepArgs {
    const u64 *cols_a, *cols_b;
    u64 *out;
    const u64 *gpow;     // [(Wa+Wb+nnext)][3]
    const u64 *twl, *twh;  // w_M^e
    u64 ca[3], cb[3];    // sum g^k e_k ,  sum g^(W+k) e'_k
    u64 z[3], zw[3];
    u64 shift;               // shift * w_M^row0 for a row window
    u64 nrows, sa, sb, so;   // rows of this launch; column strides of cols_a, cols_b, out
    int logm, Wa, Wb, nnext, lb;
    int pair_ok;             // column strides and base pointers allow 16-byte loads of row pairs
};

// numerators and denominators of one row: A = sum g^k p_k(x) - c_a, B = (prefix over the first nnext columns) g^W - c_b,
// d1 = x - z, d2 = x - zw
struct DeepRow { e3 A, B, d1, d2; };
// what follows the column sums of a row: the constant terms and the two denominators
__device__ __forceinline__ void deep_finish(const DeepArgs &a, u64 r, DeepRow &o) {
    const int W = a.Wa + a.Wb;
    if (a.nnext >= W) o.B = o.A;
    if (a.nnext > 0) o.B = e3_mul(o.B, e3_make(a.gpow[W * 3], a.gpow[W * 3 + 1], a.gpow[W * 3 + 2]));
    o.A = e3_sub(o.A, e3_make(a.ca[0], a.ca[1], a.ca[2]));
    o.B = e3_sub(o.B, e3_make(a.cb[0], a.cb[1], a.cb[2]));
    const u64 x = gl_mul(a.shift, gl_mul(a.twl[r & ((1ULL << a.lb) - 1)], a.twh[r >> a.lb]));
    o.d1 = e3_make(gl_sub(x, a.z[0]), gl_neg(a.z[1]), gl_neg(a.z[2]));
    o.d2 = e3_make(gl_sub(x, a.zw[0]), gl_neg(a.zw[1]), gl_neg(a.zw[2]));
}
__device__ __forceinline__ DeepRow deep_row(const DeepArgs &a, u64 r) {
    // three unreduced dot products; the second sum runs over the same columns with the powers shifted by W, so it is
    // g^W times the prefix of A over the first nnext columns
    gl_acc s0 = gl_acc_zero(), s1 = gl_acc_zero(), s2 = gl_acc_zero();
    DeepRow o;
    o.B = e3_make(0, 0, 0);
    const int W = a.Wa + a.Wb;
    for (int k = 0; k < W; k++) {
        if (k == a.nnext && k > 0) o.B = e3_make(gl_acc_reduce(s0), gl_acc_reduce(s1), gl_acc_reduce(s2));
        const u64 v = k < a.Wa ? a.cols_a[(u64)k * a.sa + r] : a.cols_b[(u64)(k - a.Wa) * a.sb + r];
        const u64 *g = a.gpow + k * 3;
        gl_acc_mac(s0, v, g[0]);
        gl_acc_mac(s1, v, g[1]);
        gl_acc_mac(s2, v, g[2]);
    }
    o.A = e3_make(gl_acc_reduce(s0), gl_acc_reduce(s1), gl_acc_reduce(s2));
    deep_finish(a, r, o);
    return o;
}
// rows r and r + 1 (r even, column strides even: the library's column matrices) in ONE walk over the columns, each lane reading its two
// values with one 16-byte load -- a wave's load is a 1 KiB run, where two walks with 8-byte loads at stride 16 fetched every line twice
__device__ __forceinline__ void deep_rows2(const DeepArgs &a, u64 r, DeepRow &o0, DeepRow &o1) {
    gl_acc s[2][3];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int c = 0; c < 3; c++) s[i][c] = gl_acc_zero();
    o0.B = e3_make(0, 0, 0);
    o1.B = o0.B;
    const int W = a.Wa + a.Wb;
    for (int k = 0; k < W; k++) {
        if (k == a.nnext && k > 0) {
            o0.B = e3_make(gl_acc_reduce(s[0][0]), gl_acc_reduce(s[0][1]), gl_acc_reduce(s[0][2]));
            o1.B = e3_make(gl_acc_reduce(s[1][0]), gl_acc_reduce(s[1][1]), gl_acc_reduce(s[1][2]));
        }
        const u64 *p = k < a.Wa ? a.cols_a + (u64)k * a.sa + r : a.cols_b + (u64)(k - a.Wa) * a.sb + r;
        const ulonglong2 v = *(const ulonglong2 *)p;
        const u64 *g = a.gpow + k * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gl_acc_mac(s[0][c], v.x, g[c]);
            gl_acc_mac(s[1][c], v.y, g[c]);
        }
    }
    o0.A = e3_make(gl_acc_reduce(s[0][0]), gl_acc_reduce(s[0][1]), gl_acc_reduce(s[0][2]));
    o1.A = e3_make(gl_acc_reduce(s[1][0]), gl_acc_reduce(s[1][1]), gl_acc_reduce(s[1][2]));
    deep_finish(a, r, o0);
    deep_finish(a, r + 1, o1);
}
// Two rows per lane and ONE base-field inversion (Fermat, ~85 products) for their four denominators:
// 1/d1 = d2 adj(d1 d2) / det, 1/d2 = d1 adj(d1 d2) / det per row, and 1/det_0, 1/det_1 from 1/(det_0 det_1).
__global__ void __launch_bounds__(256) deep_quotient_kernel(DeepArgs a) {
    const u64 r0 = ((u64)blockIdx.x * 256 + threadIdx.x) * 2;
    if (r0 >= a.nrows) return;
    const bool two = r0 + 1 < a.nrows;
    DeepRow q0, q1;
    if (two && a.pair_ok) deep_rows2(a, r0, q0, q1);
    else { q0 = deep_row(a, r0); q1 = two ? deep_row(a, r0 + 1) : q0; }
    u64 det0, det1;
    const e3 adj0 = e3_adj(a.nnext > 0 ? e3_mul(q0.d1, q0.d2) : q0.d1, &det0);
    const e3 adj1 = e3_adj(a.nnext > 0 ? e3_mul(q1.d1, q1.d2) : q1.d1, &det1);
    // a vanishing denominator (z on the domain: excluded by the protocol) keeps the convention 1/0 = 0 for ITS row only
    const u64 m0 = det0 ? det0 : 1, m1 = det1 ? det1 : 1;
    const u64 inv01 = gl_inv(gl_mul(m0, m1));
    const e3 pi0 = e3_scale(adj0, det0 ? gl_mul(inv01, m1) : 0), pi1 = e3_scale(adj1, det1 ? gl_mul(inv01, m0) : 0);
    e3 F0, F1;
    if (a.nnext > 0) {
        F0 = e3_add(e3_mul(q0.A, e3_mul(pi0, q0.d2)), e3_mul(q0.B, e3_mul(pi0, q0.d1)));
        F1 = e3_add(e3_mul(q1.A, e3_mul(pi1, q1.d2)), e3_mul(q1.B, e3_mul(pi1, q1.d1)));
    } else {
        F0 = e3_mul(q0.A, pi0);
        F1 = e3_mul(q1.A, pi1);
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        a.out[(u64)c * a.so + r0] = F0.c[c];
        if (two) a.out[(u64)c * a.so + r0 + 1] = F1.c[c];
    }
}

__global__ void __launch_bounds__(256) gather_rows_kernel(const u64 *cols, u64 M, int W, const u64 *idx, int nq, u64 *out) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= (u64)nq * W) return;
    const u64 q = i / W, c = i % W;
    out[i] = cols[c * M + idx[q]];
}

__global__ void __launch_bounds__(256) merkle_paths_kernel(const u64 *tree, u64 M, int depth, const u64 *idx, int nq, u64 *out) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= (u64)nq * depth * 4) return;
    const int e = (int)(i & 3);
    const u64 qd = i >> 2;
    const int d = (int)(qd % depth);
    const u64 q = qd / depth;
    u64 off = 0, cnt = M;
    for (int l = 0; l < d; l++) { off += cnt; cnt >>= 1; }
    const u64 node = (idx[q] >> d) ^ 1;
    out[i] = tree[(off + node) * 4 + e];
}

}  // namespace

static e3 to_e3(const void *pv) { const u64 *p = (const u64 *)pv; return e3_make(p[0], p[1], p[2]); }

extern "C" {

int32_t zp_poly_eval_ext(zp_ctx *ctx, const uint64_t *d_coef, int32_t logn, int32_t W, const uint64_t z[3],
                         uint64_t *h_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "poly_eval_ext");
    ZP_ARG(ctx, logn >= 0 && logn <= 32 && W >= 0, "logn/W out of range");
    ZP_ARG(ctx, (d_coef && z && h_out) || W == 0, "null pointer");
    if (W == 0) return ZP_OK;
    ZP_ARG(ctx, z[0] < GL_P && z[1] < GL_P && z[2] < GL_P, "point not canonical");
    const u64 n = 1ULL << logn;
    const u64 chunks = (n + EV_CH - 1) / EV_CH;
    std::vector<u64> tab((EV_J + 256) * 3);
    e3 zz = to_e3(z), cur = e3_make(1, 0, 0);
    for (int t = 0; t < 256; t++) { memcpy(&tab[(EV_J + t) * 3], cur.c, 24); cur = e3_mul(cur, zz); }
    e3 z256 = cur;  // z^256
    cur = e3_make(1, 0, 0);
    for (int j = 0; j < EV_J; j++) { memcpy(&tab[j * 3], cur.c, 24); cur = e3_mul(cur, z256); }
    e3 zch = cur;   // z^EV_CH
    u64 *d_tab = nullptr, *d_part = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, tab.size() + chunks * W * 3, &d_tab));
    d_part = d_tab + tab.size();
    ZP_TRY(zpi_h2d_small(ctx, d_tab, tab.data(), tab.size() * 8));
    EvalArgs a;
    a.coef = (const u64 *)d_coef; a.n = n; a.zlow = d_tab; a.zmid = d_tab + EV_J * 3; a.partial = d_part; a.W = W;
    hipLaunchKernelGGL(poly_eval_ext_kernel, dim3((unsigned)chunks, (unsigned)W), dim3(256), 0, ctx->stream, a);
    ZP_HIP(ctx, hipGetLastError());
    std::vector<u64> part(chunks * W * 3);
    if (part.size() * 8 <= ZP_SMALL_COPY) {
        ZP_TRY(zpi_d2h_small(ctx, part.data(), d_part, part.size() * 8));
    } else {
        ZP_HIP(ctx, hipMemcpyAsync(part.data(), d_part, part.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // combine the per-chunk partials on the host: sum_ch partial[ch] * (z^EV_CH)^ch   (tiny: chunks*W terms)
    for (int c = 0; c < W; c++) {
        e3 acc = e3_make(0, 0, 0);
        for (u64 ch = chunks; ch-- > 0;) {
            acc = e3_mul(acc, zch);
            acc = e3_add(acc, to_e3(&part[(ch * W + c) * 3]));
        }
        memcpy(h_out + c * 3, acc.c, 24);
    }
    return ZP_OK;
}

int32_t zp_ood_eval(zp_ctx *ctx, const uint64_t *d_cols, size_t col_stride, size_t row_stride, int32_t W, int32_t logn, uint64_t shift,
                    const uint64_t z[3], int32_t want_next, uint64_t *h_ev_z, uint64_t *h_ev_zw) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "ood_eval");
    ZP_ARG(ctx, logn >= 0 && logn <= 32 && W >= 0, "logn/W out of range");
    if (W == 0) return ZP_OK;
    ZP_ARG(ctx, d_cols && z && h_ev_z && (h_ev_zw || !want_next), "null pointer");
    ZP_ARG(ctx, z[0] < GL_P && z[1] < GL_P && z[2] < GL_P, "point not canonical");
    ZP_ARG(ctx, row_stride >= 1 && col_stride >= (((size_t)1 << logn) - 1) * row_stride + 1, "strides do not hold a column");
    if (shift == 0) shift = ctx->coset_shift;
    ZP_ARG(ctx, shift < GL_P, "shift not canonical");
    const u64 n = 1ULL << logn;
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logn, false, &pl));
    const e3 y = e3_scale(to_e3(z), gl_inv(shift));          // the columns hold P(w^i) = p(shift w^i): evaluate P at z / shift
    const u64 blocks = (n + BY_ROWS - 1) / BY_ROWS;
    const size_t W6 = (size_t)W * 6;
    // The weights depend on (n, y) only: a prover evaluates its trace columns, its stage-2 columns and (Q > 1) its quotient pieces at the same
    // point of the same domain in consecutive calls, so the last weight vector is kept (scratch 4 of this entry point: [u: 3 n][flag]) and
    // reused while nothing reallocated it; sums and partials live in scratch 5.
    u64 *d_u = nullptr, *d_sum = nullptr;
    const u64 *before = ctx->scratch[5];
    const size_t before_elems = ctx->scratch_elems[5];
    ZP_TRY(zpi_scratch(ctx, 5, 3 * (size_t)n + 1, &d_u));
    u64 *d_flag = d_u + 3 * n;
    const bool cached = before == d_u && before_elems == ctx->scratch_elems[5] && ctx->ood_valid && ctx->ood_logn == logn && ctx->ood_root32 == ctx->root32 &&
                        memcmp(ctx->ood_y, y.c, 24) == 0;
    ZP_TRY(zpi_scratch(ctx, 3, W6 + (size_t)blocks * W6, &d_sum));
    u64 *d_part = d_sum + W6;
    if (!want_next) ZP_HIP(ctx, hipMemsetAsync(d_part, 0, (size_t)blocks * W6 * 8, ctx->stream));   // components 3..5 are not written
    if (!cached) {
        ctx->ood_valid = false;
        ZP_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, ctx->stream));
        BaryWArgs wa;
        wa.u = d_u; wa.flag = (unsigned int *)d_flag; wa.twl = pl->d_twl; wa.twh = pl->d_twh; wa.lb = pl->lb; wa.n = n;
        memcpy(wa.y, y.c, 24);
        hipLaunchKernelGGL(bary_weights_kernel, dim3((unsigned)((n + 511) / 512)), dim3(256), 0, ctx->stream, wa);
        ZP_HIP(ctx, hipGetLastError());
    }
    BaryArgs a;
    a.cols = (const u64 *)d_cols; a.u = d_u; a.partial = d_part; a.cs = col_stride; a.rs = row_stride; a.n = n; a.W = W;
    a.blocks = blocks; a.groups = (W + BY_CB - 1) / BY_CB;
    ZP_ARG(ctx, ((blocks + 7) / 8) * 8 * (u64)a.groups < (1ULL << 31), "too many workgroups");
    const dim3 grid((unsigned)(((blocks + 7) / 8) * 8 * (u64)a.groups));
    if (want_next) hipLaunchKernelGGL(bary_dot_kernel<true>, grid, dim3(256), 0, ctx->stream, a);
    else hipLaunchKernelGGL(bary_dot_kernel<false>, grid, dim3(256), 0, ctx->stream, a);
    ZP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(bary_reduce_kernel, dim3((unsigned)W6), dim3(256), 0, ctx->stream, (const u64 *)d_part, d_sum, blocks, (int)W6);
    ZP_HIP(ctx, hipGetLastError());
    std::vector<u64> sums(1 + W6);
    ZP_TRY(zpi_d2h_small(ctx, sums.data() + 1, d_sum, W6 * 8));
    if (!cached) {
        ZP_TRY(zpi_d2h_small(ctx, sums.data(), d_flag, 8));
        if (sums[0] != 0) {
            ctx->err = "bad argument: the evaluation point lies on the evaluation domain";
            return ZP_ERR_ARG;
        }
        ctx->ood_valid = true;
        ctx->ood_logn = logn;
        ctx->ood_root32 = ctx->root32;
        memcpy(ctx->ood_y, y.c, 24);
    }
    // (y^n - 1) / n
    e3 yn = y;
    for (int i = 0; i < logn; i++) yn = e3_mul(yn, yn);
    const e3 fac = e3_scale(e3_make(gl_sub(yn.c[0], 1), yn.c[1], yn.c[2]), gl_inv(n % GL_P));
    for (int c = 0; c < W; c++) {
        const e3 ez = e3_mul(fac, to_e3(&sums[1 + (size_t)c * 6]));
        memcpy(h_ev_z + (size_t)c * 3, ez.c, 24);
        if (want_next) {
            const e3 ezw = e3_mul(fac, to_e3(&sums[1 + (size_t)c * 6 + 3]));
            memcpy(h_ev_zw + (size_t)c * 3, ezw.c, 24);
        }
    }
    return ZP_OK;
}

int32_t zp_deep_quotient_rows(zp_ctx *ctx, const uint64_t *d_cols_a, int32_t Wa, size_t stride_a, const uint64_t *d_cols_b, int32_t Wb,
                              size_t stride_b, int32_t logm, size_t row0, size_t nrows, int32_t n_next, const uint64_t z[3],
                              const uint64_t zw[3], const uint64_t gamma[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw,
                              uint64_t shift, uint64_t *d_out, size_t stride_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZpStage stage_(ctx, "deep_quotient");
    ZP_ARG(ctx, logm >= 0 && logm <= 32, "logm out of range");
    ZP_ARG(ctx, Wa >= 1 && Wb >= 0 && n_next >= 0 && n_next <= Wa, "bad widths");
    ZP_ARG(ctx, d_cols_a && (d_cols_b || Wb == 0) && z && zw && gamma && h_ev_z && (h_ev_zw || n_next == 0) && d_out,
           "null pointer");
    const u64 M = 1ULL << logm;
    ZP_ARG(ctx, row0 + nrows <= M && stride_a >= nrows && (Wb == 0 || stride_b >= nrows) && stride_out >= nrows, "row window / strides out of range");
    if (nrows == 0) return ZP_OK;
    if (shift == 0) shift = ctx->coset_shift;
    const int W = Wa + Wb;
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logm, false, &pl));
    std::vector<u64> gp((size_t)(W + n_next) * 3);
    e3 g = to_e3(gamma), cur = e3_make(1, 0, 0), ca = e3_make(0, 0, 0), cb = e3_make(0, 0, 0);
    for (int k = 0; k < W + n_next; k++) {
        memcpy(&gp[(size_t)k * 3], cur.c, 24);
        if (k < W) ca = e3_add(ca, e3_mul(cur, to_e3(h_ev_z + (size_t)k * 3)));
        else cb = e3_add(cb, e3_mul(cur, to_e3(h_ev_zw + (size_t)(k - W) * 3)));
        cur = e3_mul(cur, g);
    }
    u64 *d_gp = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, gp.size(), &d_gp));
    ZP_TRY(zpi_h2d_small(ctx, d_gp, gp.data(), gp.size() * 8));
    DeepArgs a;
    a.cols_a = (const u64 *)d_cols_a; a.cols_b = (const u64 *)d_cols_b; a.out = (u64 *)d_out; a.gpow = d_gp;
    a.twl = pl->d_twl; a.twh = pl->d_twh; a.lb = pl->lb;
    memcpy(a.ca, ca.c, 24); memcpy(a.cb, cb.c, 24); memcpy(a.z, z, 24); memcpy(a.zw, zw, 24);
    // x of local row j is shift * w_M^(row0 + j) = (shift * w_M^row0) * w_M^j: the window only changes the constant factor
    a.shift = gl_mul(shift, gl_pow(gl_root(ctx->root32, logm), (u64)row0));
    a.logm = logm; a.Wa = Wa; a.Wb = Wb; a.nnext = n_next;
    a.nrows = nrows; a.sa = stride_a; a.sb = stride_b; a.so = stride_out;
    a.pair_ok = (stride_a % 2 == 0 || Wa <= 1) && (stride_b % 2 == 0 || Wb <= 1) && ((uintptr_t)d_cols_a % 16 == 0) && (Wb == 0 || (uintptr_t)d_cols_b % 16 == 0);
    hipLaunchKernelGGL(deep_quotient_kernel, dim3((unsigned)((nrows + 511) / 512)), dim3(256), 0, ctx->stream, a);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

int32_t zp_deep_quotient(zp_ctx *ctx, const uint64_t *d_cols_a, int32_t Wa, const uint64_t *d_cols_b, int32_t Wb,
                         int32_t logm, int32_t n_next, const uint64_t z[3], const uint64_t zw[3],
                         const uint64_t gamma[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, uint64_t shift,
                         uint64_t *d_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, logm >= 0 && logm <= 32, "logm out of range");
    const size_t M = (size_t)1 << logm;
    return zp_deep_quotient_rows(ctx, d_cols_a, Wa, M, d_cols_b, Wb, M, logm, 0, M, n_next, z, zw, gamma, h_ev_z, h_ev_zw, shift, d_out, M);
}

int32_t zp_gather_rows(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, const uint64_t *h_idx, int32_t nq,
                       uint64_t *h_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, W >= 1 && nq >= 0 && M >= 1, "bad sizes");
    if (nq == 0) return ZP_OK;
    ZP_ARG(ctx, d_cols && h_idx && h_out, "null pointer");
    for (int i = 0; i < nq; i++) ZP_ARG(ctx, h_idx[i] < M, "row index out of range");
    u64 *d = nullptr;
    const u64 total = (u64)nq * W;
    if ((total + (u64)nq) * 8 <= ZP_SMALL_COPY) {
        // query openings are a few hundred kilobytes: indices in and rows out through the page-locked, device-visible staging buffer -- the
        // kernel reads and writes it directly: one launch, two synchronisations (round 5; before: copy kernels in and out, four synchronisations)
        void *stv = nullptr;
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ZP_TRY(zpi_pinned(ctx, (size_t)(total + (u64)nq) * 8, &stv));
        u64 *st = (u64 *)stv;
        memcpy(st, h_idx, (size_t)nq * 8);
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_cols, (u64)M, (int)W, (const u64 *)st, (int)nq, st + nq);
        ZP_HIP(ctx, hipGetLastError());
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(h_out, st + nq, (size_t)total * 8);
        return ZP_OK;
    }
    ZP_TRY(zpi_scratch(ctx, 3, (size_t)nq * (W + 1), &d));
    ZP_TRY(zpi_h2d_small(ctx, d, h_idx, (size_t)nq * 8));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const u64 *)d_cols, (u64)M, (int)W, d, (int)nq, d + nq);
    ZP_HIP(ctx, hipGetLastError());
    if (total * 8 <= ZP_SMALL_COPY) return zpi_d2h_small(ctx, h_out, d + nq, total * 8);
    ZP_HIP(ctx, hipMemcpyAsync(h_out, d + nq, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

int32_t zp_merkle_open_batch(zp_ctx *ctx, const uint64_t *d_tree, size_t M, const uint64_t *h_idx, int32_t nq,
                             uint64_t *h_paths) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    ZP_ARG(ctx, M >= 1 && (M & (M - 1)) == 0, "M must be a power of two");
    ZP_ARG(ctx, nq >= 0, "bad query count");
    int depth = 0;
    while (((size_t)1 << depth) < M) depth++;
    if (nq == 0 || depth == 0) return ZP_OK;
    ZP_ARG(ctx, d_tree && h_idx && h_paths, "null pointer");
    for (int i = 0; i < nq; i++) ZP_ARG(ctx, h_idx[i] < M, "leaf index out of range");
    u64 *d = nullptr;
    const u64 total = (u64)nq * depth * 4;
    if ((total + (u64)nq) * 8 <= ZP_SMALL_COPY) {      // as zp_gather_rows: through the staging buffer, one launch
        void *stv = nullptr;
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ZP_TRY(zpi_pinned(ctx, (size_t)(total + (u64)nq) * 8, &stv));
        u64 *st = (u64 *)stv;
        memcpy(st, h_idx, (size_t)nq * 8);
        hipLaunchKernelGGL(merkle_paths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                           (const u64 *)d_tree, (u64)M, depth, (const u64 *)st, (int)nq, st + nq);
        ZP_HIP(ctx, hipGetLastError());
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        memcpy(h_paths, st + nq, (size_t)total * 8);
        return ZP_OK;
    }
    ZP_TRY(zpi_scratch(ctx, 3, (size_t)nq + total, &d));
    ZP_TRY(zpi_h2d_small(ctx, d, h_idx, (size_t)nq * 8));
    hipLaunchKernelGGL(merkle_paths_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const u64 *)d_tree, (u64)M, depth, d, (int)nq, d + nq);
    ZP_HIP(ctx, hipGetLastError());
    if (total * 8 <= ZP_SMALL_COPY) return zpi_d2h_small(ctx, h_paths, d + nq, total * 8);
    ZP_HIP(ctx, hipMemcpyAsync(h_paths, d + nq, total * 8, hipMemcpyDeviceToHost, ctx->stream));
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZP_OK;
}

}  // extern "C"

// ---- grand product (stage-2 witness of a permutation argument) ---------------------------------------
// Z[0] = 1,  Z[i+1] = Z[i] * (a[i] + g) / (b[i] + g)   in F_{p^3};  out planes u64[3][N].
// Three launches: (1) per-lane ratios r_i and the exclusive product of each lane's 16 ratios, block scan
// in LDS, one total per block;  (2) exclusive scan of the block totals (one workgroup);  (3) offsets.
namespace {

struct GpArgs {
    const u64 *a, *b;
    u64 *out;       // [3][N]
    e3 *totals;     // [nblocks]
    u64 g[3];
    u64 n;
};
#define GP_PER 16
#define GP_BLK 256

__device__ __forceinline__ e3 gp_ratio(const GpArgs &A, u64 i) {
    const e3 num = e3_make(gl_add(A.a[i], A.g[0]), A.g[1], A.g[2]);
    const e3 den = e3_make(gl_add(A.b[i], A.g[0]), A.g[1], A.g[2]);
    return e3_mul(num, e3_inv(den));
}

__global__ void __launch_bounds__(GP_BLK) gp_local_kernel(GpArgs A) {
    __shared__ e3 sh[GP_BLK];
    const int t = threadIdx.x;
    const u64 base = ((u64)blockIdx.x * GP_BLK + t) * GP_PER;
    e3 acc = e3_make(1, 0, 0);
    // exclusive products inside the lane's run, stored unscaled; scaled by the lane/block prefix later
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
#pragma unroll
            for (int c = 0; c < 3; c++) A.out[(u64)c * A.n + i] = acc.c[c];
            acc = e3_mul(acc, gp_ratio(A, i));
        }
    }
    sh[t] = acc;
    __syncthreads();
    // inclusive Hillis-Steele scan of the lane totals
    for (int d = 1; d < GP_BLK; d <<= 1) {
        e3 v = sh[t];
        if (t >= d) v = e3_mul(sh[t - d], v);
        __syncthreads();
        sh[t] = v;
        __syncthreads();
    }
    const e3 lane_prefix = t ? sh[t - 1] : e3_make(1, 0, 0);
    if (t == GP_BLK - 1) A.totals[blockIdx.x] = sh[t];
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
            e3 v = e3_mul(lane_prefix, e3_make(A.out[i], A.out[A.n + i], A.out[2 * A.n + i]));
#pragma unroll
            for (int c = 0; c < 3; c++) A.out[(u64)c * A.n + i] = v.c[c];
        }
    }
}

// exclusive scan of the block totals, in place, one workgroup (sequential over chunks of GP_BLK)
__global__ void __launch_bounds__(GP_BLK) gp_scan_totals_kernel(e3 *totals, u64 nblocks) {
    __shared__ e3 sh[GP_BLK];
    __shared__ e3 carry;
    const int t = threadIdx.x;
    if (t == 0) carry = e3_make(1, 0, 0);
    __syncthreads();
    for (u64 base = 0; base < nblocks; base += GP_BLK) {
        const u64 i = base + t;
        sh[t] = i < nblocks ? totals[i] : e3_make(1, 0, 0);
        __syncthreads();
        for (int d = 1; d < GP_BLK; d <<= 1) {
            e3 v = sh[t];
            if (t >= d) v = e3_mul(sh[t - d], v);
            __syncthreads();
            sh[t] = v;
            __syncthreads();
        }
        const e3 excl = e3_mul(carry, t ? sh[t - 1] : e3_make(1, 0, 0));
        const e3 last = e3_mul(carry, sh[GP_BLK - 1]);
        __syncthreads();
        if (i < nblocks) totals[i] = excl;
        if (t == 0) carry = last;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(GP_BLK) gp_apply_kernel(GpArgs A) {
    const e3 off = A.totals[blockIdx.x];
    const u64 base = ((u64)blockIdx.x * GP_BLK + threadIdx.x) * GP_PER;
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
            e3 v = e3_mul(off, e3_make(A.out[i], A.out[A.n + i], A.out[2 * A.n + i]));
#pragma unroll
            for (int c = 0; c < 3; c++) A.out[(u64)c * A.n + i] = v.c[c];
        }
    }
}

// ---- LogUp lookup columns (stage-2 witness of a range-check / lookup argument)
// h1 = 1/(a+g), h2 = m/(t+g), S[0] = 0, S[i+1] = S[i] + h1[i] - h2[i]   (all in F_{p^3}; planes h1 0..2, h2 3..5, S 6..8)
struct LuArgs {
    const u64 *a, *t, *m;
    u64 *out;
    e3 *totals;
    u64 n;
    u64 g[3];
};

__global__ void __launch_bounds__(GP_BLK) lu_local_kernel(LuArgs A) {
    __shared__ e3 sh[GP_BLK];
    const int t = threadIdx.x;
    const u64 base = ((u64)blockIdx.x * GP_BLK + t) * GP_PER;
    e3 acc = e3_make(0, 0, 0);
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
            const e3 h1 = e3_inv(e3_make(gl_add(A.a[i], A.g[0]), A.g[1], A.g[2]));
            const e3 h2 = e3_scale(e3_inv(e3_make(gl_add(A.t[i], A.g[0]), A.g[1], A.g[2])), A.m[i]);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                A.out[(u64)c * A.n + i] = h1.c[c];
                A.out[(u64)(3 + c) * A.n + i] = h2.c[c];
                A.out[(u64)(6 + c) * A.n + i] = acc.c[c];
            }
            acc = e3_add(acc, e3_sub(h1, h2));
        }
    }
    sh[t] = acc;
    __syncthreads();
    for (int d = 1; d < GP_BLK; d <<= 1) {
        e3 v = sh[t];
        if (t >= d) v = e3_add(sh[t - d], v);
        __syncthreads();
        sh[t] = v;
        __syncthreads();
    }
    const e3 lane_prefix = t ? sh[t - 1] : e3_make(0, 0, 0);
    if (t == GP_BLK - 1) A.totals[blockIdx.x] = sh[t];
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
#pragma unroll
            for (int c = 0; c < 3; c++) A.out[(u64)(6 + c) * A.n + i] = gl_add(A.out[(u64)(6 + c) * A.n + i], lane_prefix.c[c]);
        }
    }
}

__global__ void __launch_bounds__(GP_BLK) lu_scan_totals_kernel(e3 *totals, u64 nblocks) {
    __shared__ e3 sh[GP_BLK];
    __shared__ e3 carry;
    const int t = threadIdx.x;
    if (t == 0) carry = e3_make(0, 0, 0);
    __syncthreads();
    for (u64 base = 0; base < nblocks; base += GP_BLK) {
        const u64 i = base + t;
        sh[t] = i < nblocks ? totals[i] : e3_make(0, 0, 0);
        __syncthreads();
        for (int d = 1; d < GP_BLK; d <<= 1) {
            e3 v = sh[t];
            if (t >= d) v = e3_add(sh[t - d], v);
            __syncthreads();
            sh[t] = v;
            __syncthreads();
        }
        const e3 excl = e3_add(carry, t ? sh[t - 1] : e3_make(0, 0, 0));
        const e3 last = e3_add(carry, sh[GP_BLK - 1]);
        __syncthreads();
        if (i < nblocks) totals[i] = excl;
        if (t == 0) carry = last;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(GP_BLK) lu_apply_kernel(LuArgs A) {
    const e3 off = A.totals[blockIdx.x];
    const u64 base = ((u64)blockIdx.x * GP_BLK + threadIdx.x) * GP_PER;
    for (int k = 0; k < GP_PER; k++) {
        const u64 i = base + k;
        if (i < A.n) {
#pragma unroll
            for (int c = 0; c < 3; c++) A.out[(u64)(6 + c) * A.n + i] = gl_add(A.out[(u64)(6 + c) * A.n + i], off.c[c]);
        }
    }
}

}  // namespace

extern "C" int32_t zp_logup_columns(zp_ctx *ctx, const uint64_t *d_a, const uint64_t *d_t, const uint64_t *d_m, size_t n,
                                    const uint64_t gamma[3], uint64_t *d_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "logup_columns");
    ZP_ARG(ctx, n >= 1, "n must be >= 1");
    ZP_ARG(ctx, d_a && d_t && d_m && gamma && d_out, "null pointer");
    ZP_ARG(ctx, gamma[0] < GL_P && gamma[1] < GL_P && gamma[2] < GL_P, "challenge not canonical");
    const u64 nblocks = (n + (u64)GP_BLK * GP_PER - 1) / ((u64)GP_BLK * GP_PER);
    u64 *scr = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, nblocks * 3 + 8, &scr));
    LuArgs A;
    A.a = (const u64 *)d_a; A.t = (const u64 *)d_t; A.m = (const u64 *)d_m; A.out = (u64 *)d_out; A.totals = (e3 *)scr; A.n = n;
    for (int i = 0; i < 3; i++) A.g[i] = gamma[i];
    hipLaunchKernelGGL(lu_local_kernel, dim3((unsigned)nblocks), dim3(GP_BLK), 0, ctx->stream, A);
    hipLaunchKernelGGL(lu_scan_totals_kernel, dim3(1), dim3(GP_BLK), 0, ctx->stream, (e3 *)scr, nblocks);
    hipLaunchKernelGGL(lu_apply_kernel, dim3((unsigned)nblocks), dim3(GP_BLK), 0, ctx->stream, A);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}


extern "C" int32_t zp_grand_product(zp_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t n, const uint64_t gamma[3],
                                    uint64_t *d_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "grand_product");
    ZP_ARG(ctx, n >= 1, "n must be >= 1");
    ZP_ARG(ctx, d_a && d_b && gamma && d_out, "null pointer");
    ZP_ARG(ctx, gamma[0] < GL_P && gamma[1] < GL_P && gamma[2] < GL_P, "challenge not canonical");
    const u64 nblocks = (n + (u64)GP_BLK * GP_PER - 1) / ((u64)GP_BLK * GP_PER);
    u64 *scr = nullptr;
    ZP_TRY(zpi_scratch(ctx, 3, nblocks * 3 + 8, &scr));
    GpArgs A;
    A.a = (const u64 *)d_a; A.b = (const u64 *)d_b; A.out = (u64 *)d_out; A.totals = (e3 *)scr; A.n = n;
    for (int i = 0; i < 3; i++) A.g[i] = gamma[i];
    hipLaunchKernelGGL(gp_local_kernel, dim3((unsigned)nblocks), dim3(GP_BLK), 0, ctx->stream, A);
    hipLaunchKernelGGL(gp_scan_totals_kernel, dim3(1), dim3(GP_BLK), 0, ctx->stream, (e3 *)scr, nblocks);
    hipLaunchKernelGGL(gp_apply_kernel, dim3((unsigned)nblocks), dim3(GP_BLK), 0, ctx->stream, A);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}

// ---- N4 through the C-ABI: interpreter of a constraint program (include/zeth_prover.h "constraint program") ----------
// A host that cannot generate and compile a kernel per AIR (the Rust host of src/prover/provider.rs:358-377 has no hipcc at
// run time) hands the AIR over as data.  lane = LDE row; the instruction stream is wave-uniform, so decoding is scalar
// work (s_load + SALU) and every operand kind is a uniform branch; the slot file lives in LDS as [slot][lane] (no bank
// conflicts: consecutive lanes, consecutive 8-byte words).  Constraint k is folded into three unreduced 160-bit
// accumulators (alpha^k planes) at its OUT instruction; one reduction and the 1/Z_H multiplication at the end.
namespace {

constexpr int QP_MAX_SLOTS = 32;   // 32 * 256 lanes * 8 B = 64 KiB of LDS per workgroup

struct QProgArgs {
    const u64 *prog;      // device copy of the blob
    const u64 *cols, *fixedc;
    const u64 *pub, *apow, *zhinv;
    const u64 *xs_lo, *xs_hi;
    u64 *out;
    u64 M, b, shift, wlast;
    u64 nrows, sc, sf, so;   // rows of this launch (a window of the M-row domain), column strides of cols / fixed / out
    int wrap;                // 1: the window is the whole domain, the next row wraps mod M; 0: rows r + b are in the buffer (halo)
    int lb, n_const, n_instr, n_slots;
    // fixed columns 2..: periodic, kept as ONE extended period each (2^(lp + logb) values) behind the two selector columns;
    // fx_tab holds (offset, index mask) per column, row0 = the domain row of local row 0
    const u64 *fixedx, *fx_tab;
    u64 row0;
};

__global__ void __launch_bounds__(256) quotient_program_kernel(QProgArgs a) {
    extern __shared__ __attribute__((aligned(16))) u64 slots[];   // [n_slots][256]
    const int tid = threadIdx.x;
    const u64 r = (u64)blockIdx.x * 256 + tid;
    const bool live = r < a.nrows;
    const u64 rr = live ? r : 0;
    const u64 rn = a.wrap ? ((rr + a.b) & (a.M - 1)) : rr + a.b;
    const u64 x = gl_mul(a.shift, gl_mul(a.xs_lo[rr & ((1ULL << a.lb) - 1)], a.xs_hi[rr >> a.lb]));
    const u64 xml = gl_sub(x, a.wlast);
    const u64 *consts = a.prog + 12, *ins = consts + a.n_const;
    gl_acc s0 = gl_acc_zero(), s1 = gl_acc_zero(), s2 = gl_acc_zero();
    int k_out = 0;
    for (int i = 0; i < a.n_instr; i++) {
        const u64 w = ins[i];                                   // uniform address: scalar load
        const u32 wl = __builtin_amdgcn_readfirstlane((u32)w), wh = __builtin_amdgcn_readfirstlane((u32)(w >> 32));
        const u64 wu = ((u64)wh << 32) | wl;
        const int op = (int)(wu & 0xFF), dst = (int)((wu >> 8) & 0xFFFF);
        u64 v[2];
#pragma unroll
        for (int o = 0; o < 2; o++) {
            const int kind = (int)((wu >> (24 + 20 * o)) & 0xF), idx = (int)((wu >> (28 + 20 * o)) & 0xFFFF);
            u64 val;
            switch (kind) {
                case 0: val = slots[idx * 256 + tid]; break;
                case 1: val = a.cols[(u64)idx * a.sc + rr]; break;
                case 2: val = a.cols[(u64)idx * a.sc + rn]; break;
                case 3:
                    if (idx < 2) val = a.fixedc[(u64)idx * a.sf + rr];
                    else val = a.fixedx[a.fx_tab[2 * (idx - 2)] + ((a.row0 + rr) & a.fx_tab[2 * (idx - 2) + 1])];
                    break;
                case 4: val = a.pub[idx]; break;
                case 5: val = consts[idx]; break;
                default: val = xml; break;
            }
            v[o] = val;
            if (op == 4) break;
        }
        if (op == 1) slots[dst * 256 + tid] = gl_add(v[0], v[1]);
        else if (op == 2) slots[dst * 256 + tid] = gl_sub(v[0], v[1]);
        else if (op == 3) slots[dst * 256 + tid] = gl_mul(v[0], v[1]);
        else {
            gl_acc_mac(s0, v[0], a.apow[3 * k_out]);
            gl_acc_mac(s1, v[0], a.apow[3 * k_out + 1]);
            gl_acc_mac(s2, v[0], a.apow[3 * k_out + 2]);
            k_out++;
        }
    }
    if (!live) return;
    const u64 zi = a.zhinv[r & (a.b - 1)];
    a.out[r] = gl_mul(gl_acc_reduce(s0), zi);
    a.out[a.so + r] = gl_mul(gl_acc_reduce(s1), zi);
    a.out[2 * a.so + r] = gl_mul(gl_acc_reduce(s2), zi);
}

}  // namespace

__global__ void __launch_bounds__(256) scatter_pairs_kernel(u64 *__restrict__ dst, const u64 *__restrict__ pairs, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[pairs[2 * i]] = pairs[2 * i + 1];
}

// ---- the fixed columns of a statement on the evaluation domain: what zp_eval_quotient takes as d_fixed ------------------
// Layout: [L_first: M][L_last: M][column 2: 2^(lp_2 + logb)][column 3: ...] -- a periodic column of period p = 2^lp is
// f(x) = g(x^(N/p)) with g the interpolant of one period, and on the coset shift * <w_M> the point x^(N/p) runs over
// shift^(N/p) * <w_(p b)>: ONE extended period (an LDE of p values with the coset shift shift^(N/p)) holds every value, row r
// reads entry r mod p b.  Nothing of size M is built for them.
extern "C" size_t zp_fixed_columns_words(const uint64_t *h_program, size_t program_words, int32_t logn, int32_t logb) {
    std::vector<ZpFixedCol> fxc;
    if (!h_program || logn < 0 || logb < 0 || logn + logb > 32 || !zpi_program_fixed_table(h_program, program_words, &fxc)) return 0;
    size_t w = (size_t)2 << (logn + logb);
    for (const ZpFixedCol &fc : fxc) {
        if (fc.lp > logn) return 0;
        w += (size_t)1 << (fc.lp + logb);
    }
    return w;
}

// only_pub: (re)build just the columns that hold public inputs, into a buffer whose other columns are already there -- the provers keep a
// statement's fixed columns per ctx and refresh the public ones per proof (a verifier AIR at the service's size: 52 + 15 of its 104 sparse
// columns and both selectors never change; round 5)
int32_t zpi_fixed_columns_build(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub, int32_t logn, int32_t logb,
                                uint64_t shift, uint64_t *d_out, size_t out_words, bool only_pub) {
    ZP_ARG(ctx, h_program && d_out && logn >= 1 && logb >= 0 && logn + logb <= 32, "bad arguments");
    std::vector<ZpFixedCol> fxc;
    ZP_ARG(ctx, zpi_program_fixed_table(h_program, program_words, &fxc), "constraint program length does not match its header");
    ZP_ARG(ctx, (u64)n_pub >= h_program[4] && (h_program[4] == 0 || h_pub), "public inputs missing");
    const size_t need = zp_fixed_columns_words(h_program, program_words, logn, logb);
    ZP_ARG(ctx, need != 0 && out_words >= need, "output buffer smaller than zp_fixed_columns_words()");
    if (shift == 0) shift = ctx->coset_shift;
    ZP_ARG(ctx, shift < GL_P, "shift not canonical");
    const size_t N = (size_t)1 << logn, M = N << logb;
    auto wanted = [&](const ZpFixedCol &fc) { return !only_pub || fc.has_pub; };
    // input staging holds the selected columns only, one after the other
    size_t in_words = only_pub ? 0 : 2 * N;
    for (const ZpFixedCol &fc : fxc)
        if (wanted(fc)) in_words += (size_t)1 << fc.lp;
    if (in_words == 0) return ZP_OK;
    // the columns are sparse (a non-periodic public-input column of a verifier AIR has N rows and ~10^3 entries): the periods are
    // built ON THE DEVICE -- zero fill, then one scatter of (index, value) pairs -- instead of N-word host vectors and their upload
    std::vector<u64> pairs;
    if (!only_pub) {
        pairs.push_back(0); pairs.push_back(1);                      // L_first[0] = 1
        pairs.push_back(N + N - 1); pairs.push_back(1);              // L_last[N - 1] = 1
    }
    {
        size_t at = only_pub ? 0 : 2 * N;
        for (const ZpFixedCol &fc : fxc) {
            if (!wanted(fc)) continue;
            for (size_t e = 0; e < fc.n_entries; e++) {
                const u64 a = h_program[fc.first_entry_word + 2 * e], v = h_program[fc.first_entry_word + 2 * e + 1];
                u64 val = v;
                if (a >> 63) {
                    val = h_pub[v];
                    ZP_ARG(ctx, val < GL_P, "public input not canonical");
                }
                pairs.push_back(at + (a & ~(1ULL << 63)));
                pairs.push_back(val);
            }
            at += (size_t)1 << fc.lp;
        }
    }
    u64 *din = nullptr;
    ZP_TRY(zpi_scratch(ctx, 4, in_words + pairs.size(), &din));
    u64 *dpairs = din + in_words;
    int32_t rc = zp_dev_zero(ctx, din, in_words * 8);
    if (rc == ZP_OK && !pairs.empty()) rc = zp_h2d(ctx, dpairs, pairs.data(), pairs.size() * 8);
    if (rc == ZP_OK && !pairs.empty()) {
        const size_t np = pairs.size() / 2;
        hipLaunchKernelGGL(scatter_pairs_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, din, (const u64 *)dpairs, np);
        if (hipGetLastError() != hipSuccess) rc = ZP_ERR_HIP;
    }
    if (rc == ZP_OK && !only_pub) rc = zpi_lde(ctx, (const u64 *)din, (u64 *)d_out, nullptr, logn, logb, 2, shift);
    size_t in_at = only_pub ? 0 : 2 * N, out_at = 2 * M;
    for (size_t k = 0; rc == ZP_OK && k < fxc.size();) {      // consecutive SELECTED columns of one period go through one LDE call
        if (!wanted(fxc[k])) { out_at += (size_t)1 << (fxc[k].lp + logb); k++; continue; }
        size_t j = k;
        while (j < fxc.size() && fxc[j].lp == fxc[k].lp && wanted(fxc[j])) j++;
        const int lp = fxc[k].lp;
        const u64 sh = gl_pow(shift, (u64)1 << (logn - lp));    // shift^(N/p)
        if (logb == 0 && sh == 1) rc = zp_d2d(ctx, d_out + out_at, (const u64 *)din + in_at, ((j - k) << lp) * 8);
        else rc = zpi_lde(ctx, (const u64 *)din + in_at, (u64 *)d_out + out_at, nullptr, lp, logb, (int)(j - k), sh);
        in_at += (j - k) << lp;
        out_at += (j - k) << (lp + logb);
        k = j;
    }
    return rc;
}

extern "C" int32_t zp_fixed_columns(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub,
                                    int32_t logn, int32_t logb, uint64_t shift, uint64_t *d_out, size_t out_words) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "fixed_columns");
    return zpi_fixed_columns_build(ctx, h_program, program_words, h_pub, n_pub, logn, logb, shift, d_out, out_words, false);
}

extern "C" int32_t zp_eval_quotient_rows(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *d_cols,
                                         size_t stride_cols, const uint64_t *d_fixed, size_t stride_fixed, int32_t logm, int32_t logb,
                                         size_t row0, size_t nrows, const uint64_t *h_pub, int32_t n_pub,
                                         const uint64_t *h_alpha_pows, const uint64_t *h_zhinv, uint64_t shift, uint64_t w_last,
                                         uint64_t *d_out, size_t stride_out);

extern "C" int32_t zp_eval_quotient(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *d_cols,
                                    const uint64_t *d_fixed, int32_t logm, int32_t logb, const uint64_t *h_pub, int32_t n_pub,
                                    const uint64_t *h_alpha_pows, const uint64_t *h_zhinv, uint64_t shift, uint64_t w_last,
                                    uint64_t *d_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, logm >= 0 && logm <= 32, "logm/logb out of range");
    const size_t M = (size_t)1 << logm;
    return zp_eval_quotient_rows(ctx, h_program, program_words, d_cols, M, d_fixed, M, logm, logb, 0, M, h_pub, n_pub, h_alpha_pows,
                                 h_zhinv, shift, w_last, d_out, M);
}

extern "C" int32_t zp_eval_quotient_rows(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *d_cols,
                                         size_t stride_cols, const uint64_t *d_fixed, size_t stride_fixed, int32_t logm, int32_t logb,
                                         size_t row0, size_t nrows, const uint64_t *h_pub, int32_t n_pub,
                                         const uint64_t *h_alpha_pows, const uint64_t *h_zhinv, uint64_t shift, uint64_t w_last,
                                         uint64_t *d_out, size_t stride_out) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "eval_quotient");
    ZP_ARG(ctx, h_program && d_cols && d_fixed && h_alpha_pows && h_zhinv && d_out, "null pointer");
    ZP_ARG(ctx, logm >= 0 && logm <= 32 && logb >= 0 && logb <= logm, "logm/logb out of range");
    ZP_ARG(ctx, program_words >= 12, "constraint program shorter than its header");
    static const unsigned char magic[8] = {'Z', 'P', 'A', 'I', 'R', '1', 0, 0};
    ZP_ARG(ctx, memcmp(h_program, magic, 8) == 0, "not a ZPAIR1 constraint program");
    const u64 width = h_program[1], width2 = h_program[2], n_fixed = h_program[3], np = h_program[4], nchal = h_program[5];
    const u64 n_const = h_program[6], n_instr = h_program[7], n_cons = h_program[8], n_slots = h_program[9], n_s2 = h_program[10];
    std::vector<ZpFixedCol> fxc;
    ZP_ARG(ctx, n_const < (1u << 16) && n_instr < (1u << 24) && n_s2 < (1u << 16) && n_fixed >= 2 && n_fixed < 4096 &&
                    zpi_program_fixed_table(h_program, program_words, &fxc), "constraint program length does not match its header");
    ZP_ARG(ctx, n_slots >= 1 && n_slots <= (u64)QP_MAX_SLOTS, "constraint program needs more slots than the interpreter has (32)");
    ZP_ARG(ctx, (u64)n_pub == np + nchal && (np + nchal == 0 || h_pub), "n_pub must equal publics + challenges of the program");
    for (const ZpFixedCol &fc : fxc) ZP_ARG(ctx, fc.lp <= logm - logb, "fixed column longer than the trace");
    ZP_ARG(ctx, shift < GL_P && w_last < GL_P, "shift / w_last not canonical");
    const uint64_t *consts = h_program + 12, *ins = consts + n_const;
    u64 outs = 0;
    for (u64 i = 0; i < n_const; i++) ZP_ARG(ctx, consts[i] < GL_P, "constant not canonical");
    for (u64 i = 0; i < n_instr; i++) {
        const u64 w = ins[i];
        const unsigned op = (unsigned)(w & 0xFF), dst = (unsigned)((w >> 8) & 0xFFFF);
        ZP_ARG(ctx, op >= 1 && op <= 4, "unknown opcode in constraint program");
        for (int o = 0; o < (op == 4 ? 1 : 2); o++) {
            const unsigned kind = (unsigned)((w >> (24 + 20 * o)) & 0xF), idx = (unsigned)((w >> (28 + 20 * o)) & 0xFFFF);
            const u64 lim = kind == 0 ? n_slots : (kind == 1 || kind == 2) ? width + width2 : kind == 3 ? n_fixed
                            : kind == 4 ? np + nchal : kind == 5 ? n_const : kind == 6 ? 1 : 0;
            ZP_ARG(ctx, idx < lim, "operand out of range in constraint program");
        }
        if (op != 4) ZP_ARG(ctx, dst < n_slots, "destination slot out of range in constraint program");
        outs += op == 4;
    }
    ZP_ARG(ctx, outs == n_cons, "OUT count does not match the header");
    for (int i = 0; i < n_pub; i++) ZP_ARG(ctx, h_pub[i] < GL_P, "public input not canonical");
    const u64 M = 1ULL << logm, b = 1ULL << logb;
    const bool whole = row0 == 0 && nrows == M;
    ZP_ARG(ctx, row0 + nrows <= M && row0 % b == 0 && nrows % b == 0, "row window must be aligned to the blow-up and inside the domain");
    ZP_ARG(ctx, stride_cols >= nrows + (whole ? 0 : b) && stride_fixed >= nrows && stride_out >= nrows,
           "strides too small (a partial window needs its blow-up halo rows behind the columns)");
    if (nrows == 0) return ZP_OK;
    NttPlan *pl;
    ZP_TRY(zpi_get_plan(ctx, logm, false, &pl));
    // one upload: program (header, constants, instructions -- NOT the sparse columns' entry tables behind them: the kernel never reads those,
    // and for a verifier AIR they are 7 MB of the blob; round 5: this upload was 16 ms of a 63 ms aggregation STARK) | pub | apow | zhinv |
    // (offset, mask) of the periodic fixed columns
    const size_t pw = 12 + (size_t)n_const + (size_t)n_instr;
    const size_t np_all = (size_t)n_pub + 1, total = pw + np_all + 3 * (size_t)n_cons + (size_t)b + 2 * fxc.size() + 1;
    std::vector<u64> h(total);
    memcpy(h.data(), h_program, pw * 8);
    for (int i = 0; i < n_pub; i++) h[pw + i] = h_pub[i];
    h[pw + n_pub] = 0;
    memcpy(h.data() + pw + np_all, h_alpha_pows, 3 * (size_t)n_cons * 8);
    memcpy(h.data() + pw + np_all + 3 * (size_t)n_cons, h_zhinv, (size_t)b * 8);
    {
        u64 *t = h.data() + pw + np_all + 3 * (size_t)n_cons + (size_t)b, off = 0;
        for (size_t k = 0; k < fxc.size(); k++) {
            const u64 len = 1ULL << (fxc[k].lp + logb);
            t[2 * k] = off;
            t[2 * k + 1] = len - 1;
            off += len;
        }
    }
    u64 *d;
    ZP_TRY(zpi_scratch(ctx, 3, total, &d));
    if (total * 8 <= ZP_SMALL_COPY) ZP_TRY(zpi_h2d_small(ctx, d, h.data(), total * 8));
    else {
        ZP_HIP(ctx, hipMemcpyAsync(d, h.data(), total * 8, hipMemcpyHostToDevice, ctx->stream));
        ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    QProgArgs a;
    a.prog = d;
    a.cols = (const u64 *)d_cols;
    a.fixedc = (const u64 *)d_fixed;
    a.pub = d + pw;
    a.apow = a.pub + np_all;
    a.zhinv = a.apow + 3 * (size_t)n_cons;
    a.fx_tab = a.zhinv + (size_t)b;
    a.fixedx = (const u64 *)d_fixed + 2 * stride_fixed;
    a.row0 = row0;
    a.xs_lo = pl->d_twl;
    a.xs_hi = pl->d_twh;
    a.out = (u64 *)d_out;
    a.M = M; a.b = b; a.wlast = w_last;
    a.shift = gl_mul(shift, gl_pow(gl_root(ctx->root32, logm), (u64)row0));   // x of local row j = (shift * w_M^row0) * w_M^j
    a.nrows = nrows; a.sc = stride_cols; a.sf = stride_fixed; a.so = stride_out; a.wrap = whole ? 1 : 0;
    a.lb = pl->lb; a.n_const = (int)n_const; a.n_instr = (int)n_instr; a.n_slots = (int)n_slots;
    hipLaunchKernelGGL(quotient_program_kernel, dim3((unsigned)((nrows + 255) / 256)), dim3(256), (size_t)n_slots * 256 * 8, ctx->stream, a);
    ZP_HIP(ctx, hipGetLastError());
    return ZP_OK;
}
