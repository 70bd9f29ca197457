// The arithmetic columns of the STARK-verifier AIR's witness (eigen_zeth_amd/stark/verifier_air.py, "Arithmetic"): what GenAggregatedProof and
// the final STARK of GenFinalProof need besides the permutation blocks (proto/prover/v1/prover.proto:115-148; client
// src/prover/provider.rs:422-503).  Per query slot and inner proof: the registers that copy the opened values, the Horner accumulator of the
// DEEP sums, the evaluation points spelled by the path bits, the interpolation / fold accumulators of every FRI layer.
//
// Host C++ walks the schedule (a few thousand field products per query: milliseconds, periods in parallel on threads) and produces one record
// per 32-row block + the rows of the fold blocks; a kernel expands the records to the 21 columns IN HBM (176 MB at 2^20 rows never cross PCIe).
// The walk states the same transitions as the constraints, link by link; stark/verifier_air.py:arith_columns is the readable reference and
// the -m "not gpu" suite compares the two through the host-only entry point (zp_verifier_arith_host needs no GPU).
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstring>
#include <thread>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int NCOL = 21;   // HR[8], ACA[3], ACB[3], XI, TPX, TAU, TPT, X, XQ, XQN
constexpr int C_ACA = 8, C_ACB = 11, C_XI = 14, C_TPX = 15, C_TAU = 16, C_TPT = 17, C_X = 18, C_XQ = 19, C_XQN = 20;
constexpr int ROWS = 32, MAXR = 16;
constexpr u64 MAGIC = 0x4854495241565A50ULL;   // "PZVARITH" little-endian
constexpr int HDR = 12, TREE_WORDS = 8 + 64, FOLD_WORDS = MAXR * 3 * 8;
// offsets inside one proof's arithmetic public inputs (stark/verifier_air.py AP_*)
constexpr int AP_G = 0, AP_CA = 24, AP_CB = 27, AP_EZA = 30, AP_EZB = 33, AP_ZETA = 36, AP_ZETAW = 39;

struct Blk { int kind, t, jl, first, last, p, sub; };   // kind: 0 idle, 1 absorb, 2 node
struct Desc {
    u64 pb, periods, k, n_proofs, T, TQ, max_w, n_open, ap_n, n_fold;
    const u64 *blk, *tree, *fold;
    Blk at(u64 b) const {
        const u64 w = blk[b];
        return Blk{(int)(w & 3), (int)((w >> 2) & 63), (int)((w >> 8) & 255), (int)((w >> 16) & 1), (int)((w >> 17) & 1), (int)((w >> 18) & 0xFFFF),
                   (int)((w >> 34) & 0xFFFF)};
    }
    const u64 *tr(int t) const { return tree + (size_t)t * TREE_WORDS; }   // w, depth, x0, xq0, lg, f, beta_off, fold_base, c[32], cq[32]
};

bool parse(const uint64_t *d, size_t words, Desc *o) {
    if (!d || words < HDR || d[0] != MAGIC) return false;
    o->pb = d[1]; o->periods = d[2]; o->k = d[3]; o->n_proofs = d[4]; o->T = d[5]; o->TQ = d[6]; o->max_w = d[7]; o->n_open = d[8]; o->ap_n = d[9];
    o->n_fold = d[10];
    if (o->pb < 1 || o->pb > (1u << 24) || o->periods < 1 || o->periods > (1u << 16) || o->T < 3 || o->T > 62 || o->TQ < 1 || o->TQ + 1 >= o->T ||
        o->n_proofs < 1 || o->n_proofs > 64 || o->max_w < 4 || o->max_w > 4096 || o->n_fold > 4096 || o->ap_n < 42 || o->ap_n > (1u << 20))
        return false;
    if (words != HDR + o->pb + o->T * TREE_WORDS + o->n_fold * FOLD_WORDS) return false;
    o->blk = (const u64 *)d + HDR;
    o->tree = o->blk + o->pb;
    o->fold = o->tree + o->T * TREE_WORDS;
    for (u64 b = 0; b < o->pb; b++) {
        const Blk k = o->at(b);
        if (k.kind > 2 || (k.kind && ((u64)k.t >= o->T || (u64)k.p >= o->n_proofs))) return false;
        if (k.kind == 2 && (u64)k.jl >= 32) return false;
    }
    for (u64 t = 0; t < o->T; t++) {
        const u64 *tr = o->tr((int)t);
        if (tr[1] < 1 || tr[1] > 32 || tr[0] < 1 || tr[0] > o->max_w) return false;
        if (t > o->TQ && (tr[5] < 1 || tr[5] > 4 || tr[7] + (3 * ((u64)1 << tr[5]) + 7) / 8 > o->n_fold || tr[6] + 3 * (((u64)1 << tr[5]) - 1) > o->ap_n)) return false;
    }
    return true;
}

inline e3 e3_at(const u64 *ap, int off) { return e3_make(ap[off], ap[off + 1], ap[off + 2]); }
inline e3 e3_neg(e3 a) { return e3_make(gl_neg(a.c[0]), gl_neg(a.c[1]), gl_neg(a.c[2])); }
inline bool e3_zero(e3 a) { return (a.c[0] | a.c[1] | a.c[2]) == 0; }

struct Out {   // per block: the 21 registers at row 0, the 8 accumulator values at the rows behind the fold rows, fold rows 1 .. R-1
    u64 *A, *E, *F;       // [nblk][21], [nblk][8], [nblk][MAXR - 1][8] (F only where R > 0; indexed by block for simplicity of the walk)
    unsigned char *R;     // [nblk]
};

// one period: blocks [per * pb, (per + 1) * pb).  Returns 0, or -10 / -11 (inconsistent opened values: no accepting witness)
int walk_period(const Desc &D, u64 per, const u64 *vals, const u64 *index, const u64 *dbit, const int64_t *blk_op, const u64 *aps, const u64 *fin,
                const Out &out) {
    const u64 pb = D.pb, TQ = D.TQ, T = D.T;
    u64 X = 0, XQ = 0, XQN = 0, XI = 0, TAU = 0;
    e3 ACA = e3_make(0, 0, 0), ACB = ACA;
    u64 HR[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    Blk prev = D.at(pb - 1);          // the previous period ends with idle blocks (the transcript tail / padding): registers are zero
    if (prev.kind != 0) return -12;
    for (u64 b = 0; b < pb; b++) {
        const u64 gb = per * pb + b;
        const Blk blk = D.at(b);
        // ---- what the link row does (stark/verifier_air.py: link_roles)
        const bool tree_end = prev.kind == 2 && prev.last;
        const bool keep = prev.kind != 0 && !tree_end;
        const bool snap = tree_end && (u64)prev.t == TQ - 1, qfin = tree_end && (u64)prev.t == TQ, frifin = tree_end && (u64)prev.t > TQ,
                   lastfin = tree_end && (u64)prev.t == T - 1;
        const bool tree_start = blk.kind != 0 && (blk.kind == 1 ? blk.jl == 0 : blk.first != 0);
        const bool leaf = blk.kind != 0 && (u64)blk.t <= TQ && (blk.kind == 1 || blk.first);
        const bool lka = prev.kind != 0 && !(leaf || qfin || frifin), lkb = prev.kind != 0 && !(snap || qfin || frifin);
        const int64_t o = blk_op[gb];
        if (blk.kind != 0 && (o < 0 || (u64)o >= D.n_open)) return -12;
        const u64 d = dbit[gb] & 1;
        const u64 *ap = aps + (size_t)blk.p * D.ap_n;
        u64 x0 = 0, xq0 = 0, cm1 = 0, cqm1 = 0;
        if (blk.kind != 0 && (u64)blk.t >= TQ) {
            const u64 *tr = D.tr(blk.t);
            if (tree_start) { x0 = tr[2]; xq0 = tr[3]; }
            if (blk.kind == 2) { cm1 = gl_sub(tr[8 + blk.jl], 1); cqm1 = gl_sub(tr[8 + 32 + blk.jl], 1); }
        }
        const u64 nXQ = gl_add(keep ? XQ : 0, tree_end ? XQN : 0);
        X = gl_mul(gl_add(keep ? X : 0, x0), d ? gl_add(1, cm1) : 1);
        XQN = gl_mul(gl_add(keep ? XQN : 0, xq0), d ? gl_add(1, cqm1) : 1);
        XQ = nXQ;
        u64 nHR[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (blk.kind == 1) {
            const u64 *v = vals + (size_t)o * D.max_w + 8 * (size_t)blk.jl;
            const u64 room = D.max_w - 8 * (u64)blk.jl;
            for (u64 i = 0; i < 8 && i < room; i++) nHR[i] = v[i];
        } else if (blk.kind == 2 && blk.first) {
            const u64 *v = vals + (size_t)o * D.max_w;
            for (int i = 0; i < 4; i++) nHR[i] = v[i];
        } else if (blk.kind == 2 && blk.last && (u64)blk.t == TQ) {
            const e3 i1 = e3_inv(e3_sub(e3_make(X, 0, 0), e3_at(ap, AP_ZETA))), i2 = e3_inv(e3_sub(e3_make(X, 0, 0), e3_at(ap, AP_ZETAW)));
            for (int c = 0; c < 3; c++) { nHR[c] = i1.c[c]; nHR[3 + c] = i2.c[c]; }
        }
        e3 nACA = lka ? ACA : e3_make(0, 0, 0), nACB = lkb ? ACB : e3_make(0, 0, 0);
        if (leaf) {
            const u64 w = D.tr(blk.t)[0];
            const u64 n = blk.kind == 1 ? (w - 8 * (u64)blk.jl < 8 ? w - 8 * (u64)blk.jl : 8) : w;
            if (n < 1 || n > 8) return -12;
            const bool first_of_group = blk.t == 0 && tree_start;
            e3 h = first_of_group ? e3_make(0, 0, 0) : e3_mul(ACA, e3_at(ap, AP_G + 3 * ((int)n - 1)));
            for (u64 i = 0; i < n; i++) {
                const e3 gi = (n - 1 - i == 0) ? e3_make(1, 0, 0) : e3_at(ap, AP_G + 3 * ((int)n - 2 - (int)i));
                h = e3_add(h, e3_scale(gi, nHR[i]));
            }
            nACA = h;
        }
        if (snap) nACB = ACA;
        if (qfin) {
            const u64 *pp = aps + (size_t)prev.p * D.ap_n;
            const e3 fa = e3_mul(e3_sub(e3_mul(e3_at(pp, AP_CA), ACA), e3_at(pp, AP_EZA)), e3_make(HR[0], HR[1], HR[2]));
            const e3 fb = e3_mul(e3_sub(e3_mul(e3_at(pp, AP_CB), ACB), e3_at(pp, AP_EZB)), e3_make(HR[3], HR[4], HR[5]));
            nACA = e3_neg(e3_add(fa, fb));
        }
        if (frifin) {
            if (!e3_zero(ACA)) return -10;
            if (lastfin) {
                const u64 slot = ((gb - 1) / pb) * D.k + (u64)prev.sub;
                const u64 *fv = fin + (slot * D.n_proofs + (u64)prev.p) * 3;
                if (fv[0] != ACB.c[0] || fv[1] != ACB.c[1] || fv[2] != ACB.c[2]) return -11;
            } else {
                nACA = e3_neg(ACB);
            }
        }
        if (!keep) {
            XI = TAU = 0;
            if (blk.kind != 0 && (u64)blk.t > TQ && tree_start) {
                const u64 *tr = D.tr(blk.t);
                u64 xl = tr[2];                               // x_l = shift_l w_lg^row: the row's bits select the path constants
                const u64 row = index[o];
                for (u64 l = 0; l < tr[1]; l++)
                    if ((row >> l) & 1) xl = gl_mul(xl, tr[8 + l]);
                XI = gl_inv(xl);
                TAU = gl_mul(XQ, XI);
            }
        }
        ACA = nACA; ACB = nACB;
        memcpy(HR, nHR, sizeof HR);
        // ---- the block's records
        u64 *A = out.A + gb * NCOL;
        memcpy(A, HR, sizeof HR);
        for (int c = 0; c < 3; c++) { A[C_ACA + c] = ACA.c[c]; A[C_ACB + c] = ACB.c[c]; }
        A[C_XI] = XI; A[C_TAU] = TAU; A[C_X] = X; A[C_XQ] = XQ; A[C_XQN] = XQN; A[C_TPX] = 0; A[C_TPT] = 0;
        u64 *E = out.E + gb * 8;
        out.R[gb] = 0;
        if (blk.kind == 1 && (u64)blk.t > TQ) {               // the fold rows of a FRI layer's absorb block
            const u64 *tr = D.tr(blk.t);
            const int R = 1 << tr[5];
            const u64 *cf = D.fold + (tr[7] + (u64)blk.jl) * FOLD_WORDS;      // [MAXR][3][8]
            u64 tpx = 1, tpt = 1;
            A[C_TPX] = 1; A[C_TPT] = 1;
            out.R[gb] = (unsigned char)R;
            u64 *F = out.F + gb * (MAXR - 1) * 8;
            for (int j = 0; j < R; j++) {
                if (j) {
                    u64 *row = F + (j - 1) * 8;
                    for (int c = 0; c < 3; c++) { row[c] = ACA.c[c]; row[3 + c] = ACB.c[c]; }
                    row[6] = tpx; row[7] = tpt;
                }
                e3 Dj = e3_make(0, 0, 0);
                for (int c = 0; c < 3; c++) {
                    u64 acc = 0;
                    for (int kk = 0; kk < 8; kk++) {
                        const u64 cv = cf[(j * 3 + c) * 8 + kk];
                        if (cv) acc = gl_add(acc, gl_mul(cv, HR[kk]));
                    }
                    Dj.c[c] = acc;
                }
                const e3 bj = j == 0 ? e3_make(1, 0, 0) : e3_at(ap, (int)tr[6] + 3 * (j - 1));
                ACA = e3_add(ACA, e3_scale(Dj, tpt));
                ACB = e3_add(ACB, e3_scale(e3_mul(bj, Dj), tpx));
                if (j < R - 1) { tpx = gl_mul(tpx, XI); tpt = gl_mul(tpt, TAU); }
            }
            for (int c = 0; c < 3; c++) { E[c] = ACA.c[c]; E[3 + c] = ACB.c[c]; }
            E[6] = tpx; E[7] = tpt;
        } else {
            for (int c = 0; c < 3; c++) { E[c] = ACA.c[c]; E[3 + c] = ACB.c[c]; }
            E[6] = 0; E[7] = 0;
        }
        prev = blk;
    }
    return 0;
}

int walk_all(const Desc &D, const u64 *vals, const u64 *index, const u64 *dbit, const int64_t *blk_op, const u64 *aps, const u64 *fin, const Out &out,
             int threads) {
    if (threads < 1) threads = (int)std::thread::hardware_concurrency();
    if (threads < 1) threads = 1;
    if (threads > 16) threads = 16;
    if ((u64)threads > D.periods) threads = (int)D.periods;
    std::atomic<u64> next{0};
    std::atomic<int> rc{0};
    auto body = [&]() {
        for (;;) {
            const u64 per = next.fetch_add(1);
            if (per >= D.periods || rc.load() != 0) return;
            const int r = walk_period(D, per, vals, index, dbit, blk_op, aps, fin, out);
            if (r != 0) rc.store(r);
        }
    };
    if (threads == 1) {
        body();
    } else {
        std::vector<std::thread> ts;
        for (int i = 0; i < threads; i++) ts.emplace_back(body);
        for (auto &t : ts) t.join();
    }
    return rc.load();
}

// lane = row: the 21 columns of one row from its block's records
__global__ void __launch_bounds__(256) arith_expand_kernel(u64 *__restrict__ out, const u64 *__restrict__ A, const u64 *__restrict__ E, const u64 *__restrict__ F,
                                                           const unsigned char *__restrict__ R, u64 N) {
    const u64 row = (u64)blockIdx.x * 256 + threadIdx.x;
    if (row >= N) return;
    const u64 gb = row >> 5;
    const int r = (int)(row & 31), Rb = R[gb];
    const u64 *a = A + gb * NCOL;
#pragma unroll
    for (int c = 0; c < 8; c++) out[(u64)c * N + row] = a[c];
    out[(u64)C_XI * N + row] = a[C_XI];
    out[(u64)C_TAU * N + row] = a[C_TAU];
    out[(u64)C_X * N + row] = a[C_X];
    out[(u64)C_XQ * N + row] = a[C_XQ];
    out[(u64)C_XQN * N + row] = a[C_XQN];
    u64 v[8];   // ACA[3], ACB[3], TPX, TPT
    if (r == 0) {
        for (int c = 0; c < 6; c++) v[c] = a[C_ACA + c];
        v[6] = a[C_TPX]; v[7] = a[C_TPT];
    } else if (r < Rb) {
        const u64 *f = F + (gb * (MAXR - 1) + (u64)(r - 1)) * 8;
        for (int c = 0; c < 8; c++) v[c] = f[c];
    } else {
        const u64 *e = E + gb * 8;
        for (int c = 0; c < 8; c++) v[c] = e[c];
    }
#pragma unroll
    for (int c = 0; c < 6; c++) out[(u64)(C_ACA + c) * N + row] = v[c];
    out[(u64)C_TPX * N + row] = v[6];
    out[(u64)C_TPT * N + row] = v[7];
}

void expand_host(u64 *out, const Out &o, u64 nblk) {
    const u64 N = nblk * ROWS;
    for (u64 gb = 0; gb < nblk; gb++) {
        const u64 *a = o.A + gb * NCOL, *e = o.E + gb * 8;
        const int Rb = o.R[gb];
        for (int r = 0; r < ROWS; r++) {
            const u64 row = gb * ROWS + r;
            for (int c = 0; c < NCOL; c++) out[(u64)c * N + row] = a[c];
            if (r == 0) continue;
            const u64 *src = r < Rb ? o.F + (gb * (MAXR - 1) + (u64)(r - 1)) * 8 : e;
            for (int c = 0; c < 6; c++) out[(u64)(C_ACA + c) * N + row] = src[c];
            out[(u64)C_TPX * N + row] = src[6];
            out[(u64)C_TPT * N + row] = src[7];
        }
    }
}

struct Records {
    std::vector<u64> A, E, F;
    std::vector<unsigned char> R;
    Out view() { return Out{A.data(), E.data(), F.data(), R.data()}; }
    explicit Records(u64 nblk) : A(nblk * NCOL), E(nblk * 8), F(nblk * (MAXR - 1) * 8), R(nblk) {}
};

}  // namespace

extern "C" {

// Host-only (no GPU, no ctx): h_out u64[21][32 * blocks].  0 = ok; ZP_ERR_ARG: malformed descriptor; -10 / -11: the opened values are
// inconsistent (a FRI layer does not hold the value the layer before claims / the last fold is not the final layer): no accepting witness.
int32_t zp_verifier_arith_host(const uint64_t *desc, size_t desc_words, const uint64_t *vals, const uint64_t *index, const uint64_t *dbit,
                               const int64_t *blk_op, const uint64_t *arith_pubs, const uint64_t *final_vals, uint64_t *h_out, int32_t threads) {
    Desc D;
    if (!parse(desc, desc_words, &D) || !vals || !index || !dbit || !blk_op || !arith_pubs || !final_vals || !h_out) return ZP_ERR_ARG;
    try {
        const u64 nblk = D.pb * D.periods;
        Records rec(nblk);
        const int rc = walk_all(D, (const u64 *)vals, (const u64 *)index, (const u64 *)dbit, blk_op, (const u64 *)arith_pubs, (const u64 *)final_vals,
                                rec.view(), threads);
        if (rc != 0) return rc == -12 ? ZP_ERR_ARG : rc;
        expand_host((u64 *)h_out, rec.view(), nblk);
        return ZP_OK;
    } catch (...) {
        return ZP_ERR_NOMEM;
    }
}

// The same columns written IN HBM: d_out u64[21][32 * blocks] (the tail of the verifier trace behind its 26 hashing columns).
int32_t zp_verifier_arith_trace(zp_ctx *ctx, const uint64_t *desc, size_t desc_words, const uint64_t *vals, const uint64_t *index, const uint64_t *dbit,
                                const int64_t *blk_op, const uint64_t *arith_pubs, const uint64_t *final_vals, uint64_t *d_out, int32_t threads) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "verifier_arith_trace");
    Desc D;
    ZP_ARG(ctx, parse(desc, desc_words, &D), "malformed arithmetic-witness descriptor");
    ZP_ARG(ctx, vals && index && dbit && blk_op && arith_pubs && final_vals && d_out, "null pointer");
    try {
        const u64 nblk = D.pb * D.periods, N = nblk * ROWS;
        Records rec(nblk);
        const int rc = walk_all(D, (const u64 *)vals, (const u64 *)index, (const u64 *)dbit, blk_op, (const u64 *)arith_pubs, (const u64 *)final_vals,
                                rec.view(), threads);
        if (rc == -12) { ctx->err = "arithmetic-witness inputs do not match the descriptor"; return ZP_ERR_ARG; }
        if (rc != 0) { ctx->err = "the opened values of an inner proof are inconsistent: no accepting witness"; return rc; }
        void *dA = nullptr, *dE = nullptr, *dF = nullptr, *dR = nullptr;
        int32_t r = zp_dev_alloc(ctx, rec.A.size() * 8, &dA);
        if (r == ZP_OK) r = zp_dev_alloc(ctx, rec.E.size() * 8, &dE);
        if (r == ZP_OK) r = zp_dev_alloc(ctx, rec.F.size() * 8, &dF);
        if (r == ZP_OK) r = zp_dev_alloc(ctx, (rec.R.size() + 7) / 8 * 8, &dR);
        if (r == ZP_OK) r = zp_h2d(ctx, dA, rec.A.data(), rec.A.size() * 8);
        if (r == ZP_OK) r = zp_h2d(ctx, dE, rec.E.data(), rec.E.size() * 8);
        if (r == ZP_OK) r = zp_h2d(ctx, dF, rec.F.data(), rec.F.size() * 8);
        if (r == ZP_OK) r = zp_h2d(ctx, dR, rec.R.data(), rec.R.size());
        if (r == ZP_OK) {
            hipLaunchKernelGGL(arith_expand_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, (u64 *)d_out, (const u64 *)dA, (const u64 *)dE,
                               (const u64 *)dF, (const unsigned char *)dR, N);
            if (hipGetLastError() != hipSuccess) { ctx->err = "arith_expand_kernel launch failed"; r = ZP_ERR_HIP; }
        }
        if (r == ZP_OK) r = zp_sync(ctx);      // the records are freed below
        if (dA) (void)zp_dev_free(ctx, dA);
        if (dE) (void)zp_dev_free(ctx, dE);
        if (dF) (void)zp_dev_free(ctx, dF);
        if (dR) (void)zp_dev_free(ctx, dR);
        return r;
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
}

}  // extern "C"
