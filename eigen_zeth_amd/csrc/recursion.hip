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
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "ctx.hpp"

namespace {

constexpr int NCOL = 21;   // HR[8], ACA[3], ACB[3], XI, TPX, TAU, TPT, X, XQ, XQN
constexpr int C_ACA = 8, C_ACB = 11, C_XI = 14, C_TPX = 15, C_TAU = 16, C_TPT = 17, C_X = 18, C_XQ = 19, C_XQN = 20;
constexpr int ROWS = 32, MAXR = 16;
constexpr u64 MAGIC = 0x4854495241565A50ULL;   // "PZVARITH" little-endian
constexpr int HDR = 26, TREE_WORDS = 8 + 64, FOLD_WORDS = MAXR * 3 * 8;
constexpr u64 PUBLICS_INLINE = 64;     // longer public-input vectors enter a transcript through their commitment (stark/prover.py)
// offsets inside one proof's arithmetic public inputs (stark/verifier_air.py AP_*)
constexpr int AP_G = 0, AP_CA = 24, AP_CB = 27, AP_EZA = 30, AP_EZB = 33, AP_ZETA = 36, AP_ZETAW = 39;

struct Blk { int kind, t, jl, first, last, p, sub; };   // kind: 0 idle, 1 absorb, 2 node
struct Desc {
    u64 pb, periods, k, n_proofs, T, TQ, max_w, n_open, ap_n, n_fold;
    // the inner proofs' shape (what the hashing part of the witness and the transcripts need)
    u64 n_queries, L, tblock0, W, W2, Wq, final_log, n_pub_inner, pow_bits, logn, logb, fri_logf, fri_final_log, root32, shift;
    const u64 *blk, *tree, *fold, *script;     // script: L words  n_in | out << 8 | first << 9 | pow << 10
    u64 n_slots() const { return k * periods; }
    u64 n_fri() const { return T - TQ - 1; }
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
    o->n_queries = d[11]; o->L = d[12]; o->tblock0 = d[13]; o->W = d[14]; o->W2 = d[15]; o->Wq = d[16]; o->final_log = d[17]; o->n_pub_inner = d[18];
    o->pow_bits = d[19]; o->logn = d[20]; o->logb = d[21]; o->fri_logf = d[22]; o->fri_final_log = d[23]; o->root32 = d[24]; o->shift = d[25];
    if (o->n_queries < 1 || o->n_queries > 4096 || o->L < 1 || o->L > (1u << 20) || o->W < 1 || o->W > 4096 || o->W2 > 4096 || o->Wq < 1 || o->Wq > 64 ||
        o->final_log > 20 || o->n_pub_inner > (1u << 24) || o->pow_bits > 64 || o->logn < 1 || o->logb < 1 || o->logn + o->logb > 32 || o->fri_logf < 1 ||
        o->fri_logf > 4 || o->root32 >= GL_P || o->shift >= GL_P)
        return false;
    if (o->pb < 1 || o->pb > (1u << 24) || o->periods < 1 || o->periods > (1u << 16) || o->T < 3 || o->T > 62 || o->TQ < 1 || o->TQ + 1 >= o->T ||
        o->n_proofs < 1 || o->n_proofs > 64 || o->max_w < 4 || o->max_w > 4096 || o->n_fold > 4096 || o->ap_n < 42 || o->ap_n > (1u << 20))
        return false;
    if (words != HDR + o->pb + o->T * TREE_WORDS + o->n_fold * FOLD_WORDS + o->L) return false;
    o->blk = (const u64 *)d + HDR;
    o->tree = o->blk + o->pb;
    o->fold = o->tree + o->T * TREE_WORDS;
    o->script = o->fold + o->n_fold * FOLD_WORDS;
    if (o->tblock0 + o->n_proofs * o->L > o->pb * o->periods) return false;
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

namespace {

// ---------------------------------------------------------------------------------------------------------------------------------------
// The WHOLE witness of the verifier AIR behind one call (zp_recursion_witness): openings hashed level-synchronously on the GPU, the inner
// transcripts replayed, the public inputs assembled, the permutation blocks traced, the arithmetic columns walked and expanded -- the trace
// u64[47][N] is assembled in HBM.  Port of stark/verifier_air.py:build_witness (which stays the readable form and the CPU checker's path).

__global__ void __launch_bounds__(256) block_fill_kernel(u64 *__restrict__ out, const u64 *__restrict__ per_block, u64 N) {
    const u64 row = (u64)blockIdx.x * 256 + threadIdx.x;
    if (row < N) out[row] = per_block[row >> 5];
}

struct DevTmp {       // a device scratch buffer that frees itself
    zp_ctx *ctx;
    void *p = nullptr;
    explicit DevTmp(zp_ctx *c) : ctx(c) {}
    int32_t alloc(size_t bytes) { return zp_dev_alloc(ctx, bytes ? bytes : 8, &p); }
    ~DevTmp() { if (p) (void)zp_dev_free(ctx, p); }
};

// one batched permutation of host states through the device
int32_t perm_batch(zp_ctx *ctx, DevTmp &scratch, u64 *states, size_t count) {
    if (!count) return ZP_OK;
    ZP_TRY(zp_h2d(ctx, scratch.p, states, count * 96));
    ZP_TRY(zp_poseidon_perm(ctx, (uint64_t *)scratch.p, count));
    return zp_d2h(ctx, states, scratch.p, count * 96);
}

// the sponge of stark/transcript.py as a PLAN (round 5): what a verifier absorbs is all in the proof's header, and which permutations run when
// depends on sizes only, so a replay first lists its steps -- a block of up to 8 values overwriting the rate, or a bare permutation when the
// protocol squeezes past a rate -- and where every squeezed value will come from; the whole chain then runs in ONE launch
// (zpi_poseidon_chains, all inner proofs side by side) and the squeezed values are read off its output.  Per step the witness needs the
// input state, the number of values its block absorbed and (when the protocol reads it) the rate after it.
struct PlanSponge {
    std::vector<u64> queue;
    std::vector<u64> blocks;                 // 8 words per step
    std::vector<unsigned char> absorbs;      // per step: 1 = the block overwrites the rate
    std::vector<int> n_in;
    std::vector<char> has_out;
    struct Fix { u64 *dst; size_t step; int off; };
    std::vector<Fix> fixes;                  // dst <- rate[off] after step
    bool have = false;
    size_t have_step = 0;
    int have_off = 0;
    size_t steps() const { return absorbs.size(); }
    void absorb(const u64 *v, size_t n) { queue.insert(queue.end(), v, v + n); have = false; }
    void step(const u64 *blk, int n, bool ab) {
        for (int i = 0; i < 8; i++) blocks.push_back(i < n ? blk[i] : 0ULL);
        absorbs.push_back(ab ? 1 : 0);
        n_in.push_back(n);
        has_out.push_back(0);
    }
    void flush() {
        const size_t nb = (queue.size() + 7) / 8;
        if (nb == 0) step(nullptr, 0, false);
        for (size_t i = 0; i < nb; i++) step(&queue[8 * i], (int)(queue.size() - 8 * i < 8 ? queue.size() - 8 * i : 8), true);
        queue.clear();
        has_out.back() = 1;
        have = true; have_step = steps() - 1; have_off = 0;
    }
    void squeeze(size_t n, u64 *out) {
        for (size_t got = 0; got < n; got++) {
            if (!queue.empty() || !have || have_off == 8) flush();
            fixes.push_back(Fix{out + got, have_step, have_off++});
        }
    }
};

struct Chal { e3 zeta, gamma; std::vector<e3> betas; };

// the layout of one inner proof's transcript stream (caller-supplied; see include/zeth_prover.h)
struct Stream {
    static constexpr int MAX_T = 62;
    const u64 *digest, *pubs, *root[MAX_T], *ev_z, *ev_zw, *final_l;
    u64 nonce;
};

bool slice_stream(const Desc &D, const u64 *s, size_t words, Stream *o) {
    const u64 Wt = D.W + D.W2, nfri = D.n_fri();
    size_t need = 4 + D.n_pub_inner + 4 * D.T + 3 * (Wt + D.Wq) + 3 * Wt + ((size_t)3 << D.final_log) + (D.pow_bits ? 1 : 0);
    if (words != need || D.T > (u64)Stream::MAX_T) return false;
    o->digest = s; s += 4;
    o->pubs = s; s += D.n_pub_inner;
    for (u64 t = 0; t <= D.TQ; t++) { o->root[t] = s; s += 4; }          // trace, [stage2], quotient
    o->ev_z = s; s += 3 * (Wt + D.Wq);
    o->ev_zw = s; s += 3 * Wt;
    for (u64 l = 0; l < nfri; l++) { o->root[D.TQ + 1 + l] = s; s += 4; }
    o->final_l = s; s += (size_t)3 << D.final_log;
    o->nonce = D.pow_bits ? *s : 0;
    return true;
}

// Goldilocks-mode commitment to a long public-input vector (stark/prover.py publics_rows: rows of 8, zero padded, a power of two >= 2 of them)
int32_t publics_digest(zp_ctx *ctx, const u64 *pubs, size_t n, u64 out4[4]) {
    size_t M = 2;
    while (M * 8 < n) M <<= 1;
    std::vector<u64> rows(M * 8, 0);
    memcpy(rows.data(), pubs, n * 8);
    DevTmp d(ctx), tree(ctx);
    ZP_TRY(d.alloc(rows.size() * 8));
    ZP_TRY(tree.alloc((2 * M - 1) * 32));
    ZP_TRY(zp_h2d(ctx, d.p, rows.data(), rows.size() * 8));
    ZP_TRY(zp_merkle_commit_rows(ctx, (const uint64_t *)d.p, M, 8, (uint64_t *)tree.p));
    return zp_d2h(ctx, out4, (const char *)tree.p + (2 * M - 2) * 32, 32);
}

// replay of one inner transcript, in two halves around the one launch that runs every proof's chain.  plan: lists the steps (the digest of a
// long public vector is the one hash made on the way).  finish: `out` = u64[steps][20] of the chain (input state, rate after) -> fills `states`
// (L x 12: the inputs of its permutation blocks), appends its section of the public inputs to `tp`, returns the challenges.  -14: the
// transcript does not give the proof's indices / the grinding nonce fails; -12: wrong shape.
struct ReplayPlan {
    PlanSponge tr;
    Chal ch;
    std::vector<u64> idx;
    u64 seed[4], tmp[8], pow_out[12];
    u64 dg[4];
};

int32_t replay_plan(zp_ctx *ctx, const Desc &D, const Stream &st, ReplayPlan *rp) {
    PlanSponge &tr = rp->tr;
    Chal *ch = &rp->ch;
    std::vector<u64> head = {D.logn, D.logb, D.W, D.W2, D.fri_logf, D.fri_final_log, D.n_queries, D.pow_bits, D.root32, D.shift,
                             st.digest[0], st.digest[1], st.digest[2], st.digest[3], D.n_pub_inner};
    if (D.n_pub_inner <= PUBLICS_INLINE) {
        head.insert(head.end(), st.pubs, st.pubs + D.n_pub_inner);
        tr.absorb(head.data(), head.size());
    } else {
        ZP_TRY(publics_digest(ctx, st.pubs, D.n_pub_inner, rp->dg));
        tr.absorb(head.data(), head.size());
        tr.absorb(rp->dg, 4);
    }
    const u64 Wt = D.W + D.W2;
    tr.absorb(st.root[0], 4);
    if (D.W2) { tr.squeeze(3, rp->tmp); tr.absorb(st.root[1], 4); }
    tr.squeeze(3, rp->tmp);                                   // alpha
    tr.absorb(st.root[D.TQ], 4);
    tr.squeeze(3, ch->zeta.c);
    tr.absorb(st.ev_z, 3 * (Wt + D.Wq));
    tr.absorb(st.ev_zw, 3 * Wt);
    tr.squeeze(3, ch->gamma.c);
    ch->betas.assign(D.n_fri(), e3_make(0, 0, 0));            // (sized first: the fix-ups point into it)
    for (u64 l = 0; l < D.n_fri(); l++) {
        tr.absorb(st.root[D.TQ + 1 + l], 4);
        tr.squeeze(3, ch->betas[l].c);
    }
    tr.absorb(st.final_l, (size_t)3 << D.final_log);
    if (D.pow_bits) {
        tr.squeeze(4, rp->seed);
        if (st.nonce >= GL_P) return -14;
        tr.absorb(&st.nonce, 1);
    }
    rp->idx.assign(D.n_queries, 0);
    tr.squeeze(D.n_queries, rp->idx.data());
    return ZP_OK;
}

// the grinding hash's input, once the chain has run (seed read off it)
void replay_pow_input(const Stream &st, ReplayPlan *rp, const u64 *out, u64 pin[12]) {
    for (const PlanSponge::Fix &f : rp->tr.fixes) *f.dst = out[f.step * 20 + 12 + f.off];
    const u64 in[12] = {rp->seed[0], rp->seed[1], rp->seed[2], rp->seed[3], st.nonce, 0, 0, 0, 0, 0, 0, 0};
    memcpy(pin, in, sizeof in);
}

int32_t replay_finish(const Desc &D, const Stream &st, const u64 *index, ReplayPlan *rp, const u64 *out, const u64 *pow_in, const u64 *pow_out,
                      std::vector<u64> &states, std::vector<u64> &tp) {
    const PlanSponge &tr = rp->tr;
    if (D.pow_bits && (pow_out[0] >> (64 - D.pow_bits))) return -14;
    const u64 mask = ((u64)1 << (D.logn + D.logb)) - 1;
    for (u64 q = 0; q < D.n_queries; q++)
        if ((rp->idx[q] & mask) != index[q]) return -14;
    const size_t nrec = tr.steps() + (D.pow_bits ? 1 : 0);
    if (nrec != D.L) return -12;
    for (u64 j = 0; j < D.L; j++) {
        const u64 w = D.script[j];
        const bool is_pow = j == tr.steps();
        const int n_in = is_pow ? 5 : tr.n_in[j];
        const bool has_out = is_pow ? true : tr.has_out[j] != 0;
        if ((u64)n_in != (w & 255) || has_out != (((w >> 8) & 1) != 0)) return -12;
        const u64 *in = is_pow ? pow_in : out + j * 20, *ro = is_pow ? pow_out : out + j * 20 + 12;
        states.insert(states.end(), in, in + 12);
        tp.insert(tp.end(), in, in + n_in);
        if (has_out) tp.insert(tp.end(), ro, ro + 8);
    }
    return ZP_OK;
}

// the arithmetic section of one inner proof's public inputs (stark/verifier_air.py: arith_publics)
void arith_publics(const Desc &D, const Stream &st, const Chal &ch, u64 *out) {
    const e3 g = e3_inv(ch.gamma);
    e3 cur = e3_make(1, 0, 0);
    for (int e = 0; e < 8; e++) { cur = e3_mul(cur, g); memcpy(out + AP_G + 3 * e, cur.c, 24); }
    const u64 Wt = D.W + D.W2, Wall = Wt + D.Wq;
    e3 eza = e3_make(0, 0, 0), ezb = eza, gk = e3_make(1, 0, 0), ca = gk, cb = gk;
    for (u64 k = 0; k < Wall + Wt; k++) {
        if (k < Wall) eza = e3_add(eza, e3_mul(gk, e3_make(st.ev_z[3 * k], st.ev_z[3 * k + 1], st.ev_z[3 * k + 2])));
        else ezb = e3_add(ezb, e3_mul(gk, e3_make(st.ev_zw[3 * (k - Wall)], st.ev_zw[3 * (k - Wall) + 1], st.ev_zw[3 * (k - Wall) + 2])));
        if (k == Wall - 1) ca = gk;
        if (k == Wall + Wt - 1) cb = gk;
        gk = e3_mul(gk, ch.gamma);
    }
    const u64 wN = gl_root(D.root32, (int)D.logn);
    memcpy(out + AP_CA, ca.c, 24); memcpy(out + AP_CB, cb.c, 24); memcpy(out + AP_EZA, eza.c, 24); memcpy(out + AP_EZB, ezb.c, 24);
    memcpy(out + AP_ZETA, ch.zeta.c, 24);
    const e3 zw = e3_scale(ch.zeta, wN);
    memcpy(out + AP_ZETAW, zw.c, 24);
    for (u64 l = 0; l < D.n_fri(); l++) {
        const u64 *tr = D.tr((int)(D.TQ + 1 + l));
        e3 b = e3_make(1, 0, 0);
        for (u64 j = 1; j < ((u64)1 << tr[5]); j++) { b = e3_mul(b, ch.betas[l]); memcpy(out + tr[6] + 3 * (j - 1), b.c, 24); }
    }
}

int32_t records_to_device(zp_ctx *ctx, Records &rec, u64 nblk, u64 *d_out) {
    const u64 N = nblk * ROWS;
    DevTmp dA(ctx), dE(ctx), dF(ctx), dR(ctx);
    ZP_TRY(dA.alloc(rec.A.size() * 8)); ZP_TRY(dE.alloc(rec.E.size() * 8)); ZP_TRY(dF.alloc(rec.F.size() * 8)); ZP_TRY(dR.alloc((rec.R.size() + 7) / 8 * 8));
    ZP_TRY(zp_h2d(ctx, dA.p, rec.A.data(), rec.A.size() * 8));
    ZP_TRY(zp_h2d(ctx, dE.p, rec.E.data(), rec.E.size() * 8));
    ZP_TRY(zp_h2d(ctx, dF.p, rec.F.data(), rec.F.size() * 8));
    ZP_TRY(zp_h2d(ctx, dR.p, rec.R.data(), rec.R.size()));
    hipLaunchKernelGGL(arith_expand_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_out, (const u64 *)dA.p, (const u64 *)dE.p, (const u64 *)dF.p,
                       (const unsigned char *)dR.p, N);
    if (hipGetLastError() != hipSuccess) { ctx->err = "arith_expand_kernel launch failed"; return ZP_ERR_HIP; }
    return zp_sync(ctx);      // the records are freed on return
}

int32_t recursion_witness(zp_ctx *ctx, const Desc &D, const uint64_t *const *h_index, const uint64_t *const *h_values, const uint64_t *const *h_paths,
                          const uint64_t *const *h_stream, const size_t *stream_words, u64 *d_trace, u64 *h_pubs, size_t pubs_words, int threads) {
    const u64 pb = D.pb, nblk = D.pb * D.periods, N = nblk * ROWS, T = D.T, NP = D.n_proofs, nq = D.n_queries, nslots = D.n_slots();
    static const bool trace_on = getenv("ZP_PROVE_TRACE") != nullptr;      // measurement aid: host-clock ms since entry at the stage boundaries
    const auto t_entry = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[recursion witness] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
    };
    // ---- the transcript script fixes the size of a proof's transcript section
    u64 tp_per = 0;
    for (u64 j = 0; j < D.L; j++) tp_per += (D.script[j] & 255) + (((D.script[j] >> 8) & 1) ? 8 : 0);
    const u64 n_merkle = NP * T * 4 + nslots * NP * T, n_pub = n_merkle + NP * tp_per + NP * D.ap_n + nslots * NP * 3;
    ZP_ARG(ctx, pubs_words == n_pub, "public-input buffer does not have the size the descriptor dictates");
    std::vector<Stream> st(NP);
    for (u64 p = 0; p < NP; p++) ZP_ARG(ctx, h_index[p] && h_values[p] && h_paths[p] && h_stream[p] && slice_stream(D, (const u64 *)h_stream[p], stream_words[p], &st[p]), "bad transcript stream");
    // offsets of the per-tree blocks inside a proof's values / paths arrays (layout of zp_proof_queries_parse)
    std::vector<u64> voff(T), poff(T), wv(T), dv(T);
    {
        u64 vo = 0, po = 0;
        for (u64 t = 0; t < T; t++) { wv[t] = D.tr((int)t)[0]; dv[t] = D.tr((int)t)[1]; voff[t] = vo; poff[t] = po; vo += nq * wv[t]; po += nq * dv[t] * 4; }
    }
    // ---- the opening table: one row per opening of the schedule, in block order (stark/verifier_air.py: _opening_table)
    struct Op { u64 b0, na, nd, t, p, q; };
    std::vector<Op> ops;
    ops.reserve(D.n_open);
    std::vector<int64_t> blk_op(nblk, -1);
    for (u64 per = 0; per < D.periods; per++)
        for (u64 b = 0; b < pb;) {
            const Blk k = D.at(b);
            const bool start = k.kind != 0 && (k.kind == 1 ? k.jl == 0 : k.first != 0);
            if (!start) { b++; continue; }
            const u64 w = wv[k.t], na = w <= 4 ? 0 : (w + 7) / 8, nd = dv[k.t];
            if (b + na + nd > pb) return ZP_ERR_ARG;
            for (u64 i = 0; i < na + nd; i++) blk_op[per * pb + b + i] = (int64_t)ops.size();
            ops.push_back(Op{per * pb + b, na, nd, (u64)k.t, (u64)k.p, (per * D.k + (u64)k.sub) % nq});
            b += na + nd;
        }
    ZP_ARG(ctx, ops.size() == D.n_open, "descriptor and schedule disagree on the number of openings");
    const u64 no = ops.size(), mw = D.max_w;
    std::vector<u64> vals(no * mw, 0), index(no), dbit(nblk, 0), idxv(nblk, 0), digest(no * 4);
    // the siblings of every opening, back to back, and the (b0, na, nd) table: what openings_walk_kernel reads
    std::vector<u64> opt(no * 3), sib_off(no + 1, 0);
    for (u64 o = 0; o < no; o++) sib_off[o + 1] = sib_off[o] + ops[o].nd * 4;
    std::vector<u64> sib(sib_off[no] ? sib_off[no] : 1);
    for (u64 o = 0; o < no; o++) {
        const Op &op = ops[o];
        memcpy(&vals[o * mw], (const u64 *)h_values[op.p] + voff[op.t] + op.q * wv[op.t], wv[op.t] * 8);
        index[o] = ((const u64 *)h_index[op.p])[op.q] & (((u64)1 << op.nd) - 1);
        opt[3 * o] = op.b0; opt[3 * o + 1] = op.na; opt[3 * o + 2] = op.nd;
        if (op.nd) memcpy(&sib[sib_off[o]], (const u64 *)h_paths[op.p] + poff[op.t] + op.q * op.nd * 4, op.nd * 32);
        for (u64 lv = 0; lv < op.nd; lv++) {
            const u64 blk = op.b0 + op.na + lv;
            dbit[blk] = (index[o] >> lv) & 1;
            idxv[blk] = index[o] & (((u64)2 << lv) - 1);
        }
    }
    mark("opening table");
    // ---- leaf hashes and paths ON THE DEVICE, one launch (round 5; rounds 3-4: one batched permutation and one host round trip per absorb block
    //      and per tree level, ~30 of them): every opening's workgroup writes the 12 words entering each of its permutations straight into the
    //      block-input buffer zp_poseidon_trace reads below, and hands back its last digest -- which must be the root
    DevTmp scratch(ctx), d_in(ctx), d_opn(ctx);
    ZP_TRY(scratch.alloc(NP * 96));
    ZP_TRY(d_in.alloc(nblk * 96));
    ZP_TRY(zp_dev_zero(ctx, d_in.p, nblk * 96));            // idle blocks permute the zero state
    {
        // one upload: [op table | index | sibling offsets | values | siblings], digests behind them
        const size_t o_idx = opt.size(), o_so = o_idx + no, o_val = o_so + no, o_sib = o_val + vals.size(), o_dg = o_sib + sib.size(), words = o_dg + no * 4;
        std::vector<u64> up(o_dg);
        memcpy(&up[0], opt.data(), opt.size() * 8);
        memcpy(&up[o_idx], index.data(), no * 8);
        memcpy(&up[o_so], sib_off.data(), no * 8);
        memcpy(&up[o_val], vals.data(), vals.size() * 8);
        memcpy(&up[o_sib], sib.data(), sib.size() * 8);
        ZP_TRY(d_opn.alloc(words * 8));
        u64 *d = (u64 *)d_opn.p;
        ZP_TRY(zp_h2d(ctx, d, up.data(), up.size() * 8));
        ZP_TRY(zpi_poseidon_openings_walk(ctx, d, d + o_val, mw, d + o_idx, d + o_sib, d + o_so, no, (u64 *)d_in.p, d + o_dg));
        ZP_TRY(zp_d2h(ctx, digest.data(), d + o_dg, no * 32));
    }
    for (u64 o = 0; o < no; o++)
        if (memcmp(&digest[o * 4], st[ops[o].p].root[ops[o].t], 32) != 0) {
            ctx->err = "an opening of an inner proof does not hash to its root: no accepting witness";
            return -13;
        }
    mark("leaf hashes + paths (device)");
    // ---- public inputs: roots | indices | transcripts | arithmetic constants | final-layer values
    u64 *pub = h_pubs;
    for (u64 p = 0; p < NP; p++)
        for (u64 t = 0; t < T; t++) { memcpy(pub, st[p].root[t], 32); pub += 4; }
    for (u64 g = 0; g < nslots; g++)
        for (u64 p = 0; p < NP; p++)
            for (u64 t = 0; t < T; t++) *pub++ = ((const u64 *)h_index[p])[g % nq] & (((u64)1 << dv[t]) - 1);
    std::vector<u64> aps(NP * D.ap_n, 0), fin(nslots * NP * 3);
    // ---- transcripts: every inner proof's whole sponge chain in ONE launch, the grinding hashes in a second one (rounds 3-4: a host round
    //      trip per transcript step, ~12 per proof)
    std::vector<ReplayPlan> plans(NP);
    auto replay_error = [&](int32_t r) {
        if (r == -14) ctx->err = "the transcript of an inner proof does not give its query indices or its grinding nonce fails: no accepting witness";
        if (r == -12) ctx->err = "an inner proof does not have the shape the verifier AIR was built for";
        return r;
    };
    std::vector<unsigned int> first(NP + 1, 0);
    for (u64 p = 0; p < NP; p++) {
        const int32_t r = replay_plan(ctx, D, st[p], &plans[p]);
        if (r != ZP_OK) return replay_error(r);
        first[p + 1] = first[p] + (unsigned int)plans[p].tr.steps();
    }
    const size_t nsteps = first[NP];
    std::vector<u64> chain_out(nsteps * 20), pow_io(NP * 12, 0), pow_in(NP * 12, 0);
    {
        // one upload: [blocks (8 words per step) | first (NP + 1 u32, padded) | absorb flags (bytes, padded)], the output behind them
        const size_t w_first = (NP + 2) / 2, w_abs = (nsteps + 7) / 8, o_first = nsteps * 8, o_abs = o_first + w_first, o_out = o_abs + w_abs;
        std::vector<u64> up(o_out, 0);
        unsigned char *ab = (unsigned char *)&up[o_abs];
        for (u64 p = 0; p < NP; p++) {
            memcpy(&up[(size_t)first[p] * 8], plans[p].tr.blocks.data(), plans[p].tr.blocks.size() * 8);
            memcpy(ab + first[p], plans[p].tr.absorbs.data(), plans[p].tr.absorbs.size());
        }
        memcpy(&up[o_first], first.data(), (NP + 1) * 4);
        DevTmp d_ch(ctx);
        ZP_TRY(d_ch.alloc((o_out + nsteps * 20) * 8));
        u64 *d = (u64 *)d_ch.p;
        ZP_TRY(zp_h2d(ctx, d, up.data(), up.size() * 8));
        ZP_TRY(zpi_poseidon_chains(ctx, d, (const unsigned char *)(d + o_abs), (const unsigned int *)(d + o_first), (int)NP, d + o_out));
        ZP_TRY(zp_d2h(ctx, chain_out.data(), d + o_out, chain_out.size() * 8));
    }
    for (u64 p = 0; p < NP; p++) replay_pow_input(st[p], &plans[p], &chain_out[(size_t)first[p] * 20], &pow_in[p * 12]);
    if (D.pow_bits) {
        pow_io = pow_in;
        ZP_TRY(perm_batch(ctx, scratch, pow_io.data(), NP));
    }
    std::vector<u64> tstates;                         // the transcript blocks' input states, proof after proof: one upload into the block-input buffer
    for (u64 p = 0; p < NP; p++) {
        std::vector<u64> states, tp;
        const int32_t r = replay_finish(D, st[p], (const u64 *)h_index[p], &plans[p], &chain_out[(size_t)first[p] * 20], &pow_in[p * 12], &pow_io[p * 12], states, tp);
        if (r != ZP_OK) return replay_error(r);
        if (tp.size() != tp_per || states.size() != D.L * 12) { ctx->err = "transcript section of the wrong size"; return ZP_ERR_INTERNAL; }
        tstates.insert(tstates.end(), states.begin(), states.end());
        memcpy(pub, tp.data(), tp.size() * 8);
        pub += tp.size();
        arith_publics(D, st[p], plans[p].ch, &aps[p * D.ap_n]);
    }
    ZP_TRY(zp_h2d(ctx, (u64 *)d_in.p + D.tblock0 * 12, tstates.data(), tstates.size() * 8));
    memcpy(pub, aps.data(), aps.size() * 8);
    pub += aps.size();
    const u64 fmask = ((u64)1 << D.final_log) - 1, fl = (u64)1 << D.final_log;
    for (u64 g = 0; g < nslots; g++)
        for (u64 p = 0; p < NP; p++) {
            const u64 pos = ((const u64 *)h_index[p])[g % nq] & fmask;
            for (int c = 0; c < 3; c++) fin[(g * NP + p) * 3 + c] = st[p].final_l[c * fl + pos];
        }
    memcpy(pub, fin.data(), fin.size() * 8);
    mark("transcripts replayed");
    // ---- the trace in HBM: 24 permutation columns, direction bit, index -- launched first, so that the kernels run while the host walks the
    //      arithmetic (round 5) -- then the 21 arithmetic columns
    DevTmp d_pb(ctx);
    ZP_TRY(d_pb.alloc(2 * nblk * 8));
    ZP_TRY(zp_h2d(ctx, d_pb.p, dbit.data(), nblk * 8));
    ZP_TRY(zp_h2d(ctx, (u64 *)d_pb.p + nblk, idxv.data(), nblk * 8));
    ZP_TRY(zp_poseidon_trace(ctx, (const uint64_t *)d_in.p, nblk, (uint64_t *)d_trace, (uint64_t *)(d_trace + 12 * N), N));
    for (int c = 0; c < 2; c++) {
        hipLaunchKernelGGL(block_fill_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, d_trace + (24 + c) * N, (const u64 *)d_pb.p + c * nblk, N);
        if (hipGetLastError() != hipSuccess) { ctx->err = "block_fill_kernel launch failed"; return ZP_ERR_HIP; }
    }
    Records rec(nblk);
    const int rc = walk_all(D, vals.data(), index.data(), dbit.data(), blk_op.data(), aps.data(), fin.data(), rec.view(), threads);
    if (rc != 0) (void)zp_sync(ctx);                       // the kernels above read buffers that die with this frame
    if (rc == -12) { ctx->err = "arithmetic-witness inputs do not match the descriptor"; return ZP_ERR_ARG; }
    if (rc != 0) { ctx->err = "the opened values of an inner proof are inconsistent: no accepting witness"; return rc; }
    mark("arithmetic columns (host) beside the permutation columns (device)");
    const int32_t rrc = records_to_device(ctx, rec, nblk, d_trace + 26 * N);
    mark("trace assembled in HBM");
    return rrc;
}

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
        const u64 nblk = D.pb * D.periods;
        Records rec(nblk);
        const int rc = walk_all(D, (const u64 *)vals, (const u64 *)index, (const u64 *)dbit, blk_op, (const u64 *)arith_pubs, (const u64 *)final_vals,
                                rec.view(), threads);
        if (rc == -12) { ctx->err = "arithmetic-witness inputs do not match the descriptor"; return ZP_ERR_ARG; }
        if (rc != 0) { ctx->err = "the opened values of an inner proof are inconsistent: no accepting witness"; return rc; }
        return records_to_device(ctx, rec, nblk, (u64 *)d_out);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
}

// The whole witness of the verifier AIR (see include/zeth_prover.h).
int32_t zp_recursion_witness(zp_ctx *ctx, const uint64_t *desc, size_t desc_words, const uint64_t *const *h_index, const uint64_t *const *h_values,
                             const uint64_t *const *h_paths, const uint64_t *const *h_stream, const size_t *stream_words, uint64_t *d_trace,
                             uint64_t *h_pubs, size_t pubs_words, int32_t threads) {
    if (!ctx) return ZP_ERR_ARG;
    ZpStage stage_(ctx, "recursion_witness");
    Desc D;
    ZP_ARG(ctx, parse(desc, desc_words, &D), "malformed witness descriptor");
    ZP_ARG(ctx, h_index && h_values && h_paths && h_stream && stream_words && d_trace && h_pubs, "null pointer");
    try {
        return recursion_witness(ctx, D, h_index, h_values, h_paths, h_stream, stream_words, (u64 *)d_trace, (u64 *)h_pubs, pubs_words, threads);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
}

// number of public inputs of a proof over the verifier AIR the descriptor belongs to (0: malformed descriptor)
size_t zp_recursion_publics_words(const uint64_t *desc, size_t desc_words) {
    Desc D;
    if (!parse(desc, desc_words, &D)) return 0;
    u64 tp_per = 0;
    for (u64 j = 0; j < D.L; j++) tp_per += (D.script[j] & 255) + (((D.script[j] >> 8) & 1) ? 8 : 0);
    return (size_t)(D.n_proofs * D.T * 4 + D.n_slots() * D.n_proofs * D.T + D.n_proofs * tp_per + D.n_proofs * D.ap_n + D.n_slots() * D.n_proofs * 3);
}

}  // extern "C"
