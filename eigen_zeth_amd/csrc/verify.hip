// The out-of-domain side of a constraint program, on the host: what a VERIFIER does with a statement at the point zeta.
// GenFinalProof (proto/prover/v1/prover.proto:130-148; src/prover/provider.rs:472-503) wraps an aggregated proof the CLIENT hands in: before
// the service spends a final STARK and a Groth16 proof on it, it checks natively what the final STARK's witness does not cover -- the
// aggregation STARK's constraint identity at its out-of-domain point (eigen_zeth_amd/stark/verifier.py; round-4 advisor item: "a
// pairing-valid final proof can be produced over an aggregated proof whose arithmetic is false").  The verifier AIR of an aggregation has
// ~10^2 fixed columns with ~4 x 10^5 sparse entries: their values at zeta are sums over the entries with one F_{p^3} inversion each --
// seconds in Python, milliseconds here (one shared inversion per column by Montgomery's trick, columns spread over threads).
// No reference counterpart (the prover behind the gRPC boundary is external); the checker's own statement of the same: oracle/air_program.py
// (fixed_eval_ext, evaluate_ext) -- the tests compare the two.
#include <cstring>
#include <thread>
#include <vector>

#include "ctx.hpp"

namespace {

inline e3 e3_base(u64 v) { return e3_make(v, 0, 0); }
inline e3 e3_pow2k(e3 a, int k) { for (int i = 0; i < k; i++) a = e3_mul(a, a); return a; }

// g(y) (y^p - 1) form of one sparse periodic column at zeta: sum_e v_e w_p^pos / (p (y - w_p^pos)) * (y^p - 1), y = zeta^(N / p)
bool fixed_col_at(const uint64_t *prog, const ZpFixedCol &fc, const uint64_t *pubs, int logn, u64 root32, const e3 &zeta, const e3 &zh, e3 *out) {
    if (fc.lp > logn) return false;
    const e3 y = e3_pow2k(zeta, logn - fc.lp);
    const u64 wp = fc.lp ? gl_root(root32, fc.lp) : 1;
    const u64 pinv = gl_inv((1ULL << fc.lp) % GL_P);
    // w_p^pos from a two-level table (2 x 2^(lp/2) entries): one product per entry instead of a 64-step power
    const int lb = (fc.lp + 1) / 2;
    std::vector<u64> lo((size_t)1 << lb), hi((size_t)1 << (fc.lp - lb));
    lo[0] = 1;
    for (size_t i = 1; i < lo.size(); i++) lo[i] = gl_mul(lo[i - 1], wp);
    const u64 wl = gl_mul(lo.back(), wp);
    hi[0] = 1;
    for (size_t i = 1; i < hi.size(); i++) hi[i] = gl_mul(hi[i - 1], wl);
    std::vector<e3> den, pre;
    std::vector<u64> coef;
    den.reserve(fc.n_entries); coef.reserve(fc.n_entries);
    for (size_t e = 0; e < fc.n_entries; e++) {
        const u64 a = prog[fc.first_entry_word + 2 * e], v = prog[fc.first_entry_word + 2 * e + 1];
        const u64 val = (a >> 63) ? pubs[v] % GL_P : v;
        if (!val) continue;
        const u64 pos = a & ~(1ULL << 63);
        const u64 wj = gl_mul(lo[pos & (((u64)1 << lb) - 1)], hi[pos >> lb]);
        coef.push_back(gl_mul(gl_mul(val, wj), pinv));
        den.push_back(e3_make(gl_sub(y.c[0], wj), y.c[1], y.c[2]));
    }
    e3 run = e3_base(1);
    pre.resize(den.size());
    for (size_t i = 0; i < den.size(); i++) { pre[i] = run; run = e3_mul(run, den[i]); }
    u64 det;
    const e3 adj = e3_adj(run, &det);
    if (!den.empty() && det == 0) return false;              // zeta on the domain
    e3 inv = den.empty() ? e3_base(1) : e3_scale(adj, gl_inv(det));
    e3 acc = e3_base(0);
    for (size_t i = den.size(); i-- > 0;) {
        const e3 dinv = e3_mul(inv, pre[i]);
        inv = e3_mul(inv, den[i]);
        acc = e3_add(acc, e3_scale(dinv, coef[i]));
    }
    *out = e3_mul(acc, zh);
    return true;
}

}  // namespace

// The table of sparse periodic fixed columns behind a program's stage-2 table (layout: stark/air.py compile_program): validates the whole-blob
// length and every entry; fills `cols`.  Lives in this host-only translation unit so that everything that PARSES a program blob builds
// under the host sanitizers (tests/test_verify_fuzz.py).
bool zpi_program_fixed_table(const uint64_t *h_program, size_t program_words, std::vector<ZpFixedCol> *cols) {
    if (program_words < 12) return false;
    const u64 n_fixed = h_program[3], n_pub = h_program[4], n_const = h_program[6], n_instr = h_program[7], n_s2 = h_program[10];
    // EVERY count of the header is bounded before it enters a sum or sizes anything: n_pub + n_chal wraps for a blob that says n_pub = 2^64 - 3,
    // and a public-input entry index is checked against n_pub alone (round-5 advisor item; this file is the sanitizer / fuzz surface for blobs)
    if (n_fixed < 2 || n_fixed > 4096 || n_const > (1u << 16) || n_instr > (1u << 24) || n_s2 > (1u << 16) || n_pub > (1u << 24) ||
        h_program[5] > (1u << 24) || h_program[8] > (1u << 24))
        return false;
    size_t at = 12 + (size_t)n_const + (size_t)n_instr + 4 * (size_t)n_s2;
    if (cols) cols->clear();
    for (u64 k = 2; k < n_fixed; k++) {
        if (at >= program_words) return false;
        const u64 hd = h_program[at];
        ZpFixedCol fc;
        fc.lp = (int)(hd & 0xFF);
        fc.n_entries = (size_t)(hd >> 8);
        fc.first_entry_word = at + 1;
        fc.has_pub = false;
        if (fc.lp > 32 || fc.n_entries > ((size_t)1 << fc.lp) || at + 1 + 2 * fc.n_entries > program_words) return false;
        for (size_t e = 0; e < fc.n_entries; e++) {
            const u64 a = h_program[at + 1 + 2 * e], v = h_program[at + 2 + 2 * e];
            const bool is_pub = (a >> 63) != 0;
            if ((a & ~(1ULL << 63)) >= (1ULL << fc.lp)) return false;
            if (is_pub ? v >= n_pub : v >= GL_P) return false;
            fc.has_pub |= is_pub;
        }
        at += 1 + 2 * fc.n_entries;
        if (cols) cols->push_back(fc);
    }
    return at == program_words;
}

extern "C" {

// Values of the K constraints of a program at the out-of-domain point of a proof over a trace of 2^logn rows.
//   h_pubchal u64[n_pubchal]: the public inputs, then the stage-2 challenge components (what the prover's interpreter reads as K_PUB)
//   zeta[3]; h_ev_z / h_ev_zw u64[n_cols][3]: the committed columns' evaluations at zeta / zeta w (n_cols must be the program's W + W2)
//   h_out u64[n_out][3]: constraint k at zeta (n_out must be the program's K) (numerators: the caller combines them with its alpha powers and compares with q(zeta) Z_H(zeta))
// Fixed columns: 0 / 1 the first-row / last-row selectors (Lagrange basis polynomials), then the sparse periodic columns; "x - last" is
// zeta - w^(N-1).  ZP_ERR_ARG: malformed program, non-canonical input, zeta on the trace domain.  threads <= 0: one per core, at most 16.
static int32_t program_eval_ext_impl(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn, uint64_t root32,
                                     const uint64_t zeta3[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, int32_t n_cols, uint64_t *h_out, int32_t n_out,
                                     int32_t threads, uint64_t *h_fixed_out, int32_t n_fixed_out) {
    const bool only_fixed = h_fixed_out != nullptr;
    try {
        if (!h_program || !zeta3 || program_words < 12 || logn < 1 || logn > 32 || n_pubchal < 0 || (n_pubchal && !h_pubchal)) return ZP_ERR_ARG;
        if (!only_fixed && (!h_ev_z || !h_ev_zw || !h_out)) return ZP_ERR_ARG;
        static const unsigned char magic[8] = {'Z', 'P', 'A', 'I', 'R', '1', 0, 0};
        if (memcmp(h_program, magic, 8) != 0) return ZP_ERR_ARG;
        const size_t W = h_program[1], W2 = h_program[2], n_fixed = h_program[3], n_pub = h_program[4], n_chal = h_program[5], n_const = h_program[6],
                     n_instr = h_program[7], K = h_program[8], n_slots = h_program[9];
        std::vector<ZpFixedCol> fxc;
        if (!zpi_program_fixed_table(h_program, program_words, &fxc) || W < 1 || W >= 4096 || W2 >= 4096 || n_slots > (1u << 16) || K < 1 ||
            (size_t)n_pubchal != n_pub + n_chal || root32 == 0 || root32 >= GL_P)
            return ZP_ERR_ARG;
        // the caller's arrays are sized by ITS idea of the statement: they must be the program's (a blob with another width or constraint count
        // would be read / written past them)
        if (!only_fixed && (n_cols < 0 || (size_t)n_cols != W + W2 || n_out < 0 || (size_t)n_out != K)) return ZP_ERR_ARG;
        if (only_fixed && (n_fixed_out < 0 || (size_t)n_fixed_out != n_fixed)) return ZP_ERR_ARG;
        for (int i = 0; i < 3; i++)
            if (zeta3[i] >= GL_P) return ZP_ERR_ARG;
        for (int i = 0; i < n_pubchal; i++)
            if (h_pubchal[i] >= GL_P) return ZP_ERR_ARG;
        const size_t Wt = W + W2;
        for (size_t i = 0; !only_fixed && i < Wt * 3; i++)
            if (h_ev_z[i] >= GL_P || h_ev_zw[i] >= GL_P) return ZP_ERR_ARG;
        const e3 zeta = e3_make(zeta3[0], zeta3[1], zeta3[2]);
        const u64 N = 1ULL << logn, wN = gl_root(root32, logn), wlast = gl_pow(wN, N - 1), ninv = gl_inv(N % GL_P);
        const e3 zN = e3_pow2k(zeta, logn), zh = e3_make(gl_sub(zN.c[0], 1), zN.c[1], zN.c[2]);
        std::vector<e3> fixed(n_fixed);
        {
            u64 d0, d1;
            const e3 a0 = e3_adj(e3_make(gl_sub(zeta.c[0], 1), zeta.c[1], zeta.c[2]), &d0), a1 = e3_adj(e3_make(gl_sub(zeta.c[0], wlast), zeta.c[1], zeta.c[2]), &d1);
            if (d0 == 0 || d1 == 0) return ZP_ERR_ARG;
            fixed[0] = e3_mul(e3_scale(zh, ninv), e3_scale(a0, gl_inv(d0)));
            fixed[1] = e3_mul(e3_scale(zh, gl_mul(ninv, wlast)), e3_scale(a1, gl_inv(d1)));
        }
        // the sparse columns over threads (columns differ a lot in length: hand them out one at a time)
        unsigned nt = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
        nt = nt < 1 ? 1 : nt > 16 ? 16 : nt;
        if (fxc.size() < 8) nt = 1;
        std::vector<int> bad(nt, 0);
        auto work = [&](unsigned t) noexcept {
            try {
                for (size_t k = t; k < fxc.size(); k += nt)
                    if (!fixed_col_at(h_program, fxc[k], h_pubchal, logn, root32, zeta, zh, &fixed[2 + k])) { bad[t] = 1; return; }
            } catch (...) { bad[t] = 2; }
        };
        {
            std::vector<std::thread> th;
            unsigned started = 0;
            try { for (unsigned t = 1; t < nt; t++) { th.emplace_back(work, t); started = t; } } catch (...) {}
            work(0);
            for (unsigned t = started + 1; t < nt; t++) work(t);      // threads that could not be started: their share runs here
            for (auto &x : th) x.join();
        }
        for (int b : bad)
            if (b) return b == 2 ? ZP_ERR_NOMEM : ZP_ERR_ARG;
        if (only_fixed) {
            for (size_t k = 0; k < n_fixed; k++) memcpy(h_fixed_out + 3 * k, fixed[k].c, 24);
            return ZP_OK;
        }
        const e3 xml = e3_make(gl_sub(zeta.c[0], wlast), zeta.c[1], zeta.c[2]);
        // the three-address code in F_{p^3} (layout: stark/air.py compile_program; the prover's interpreter: csrc/stark.hip quotient_program_kernel)
        const uint64_t *consts = h_program + 12, *ins = consts + n_const;
        std::vector<e3> slot(n_slots ? n_slots : 1, e3_base(0));
        size_t k_out = 0;
        bool ok = true;
        auto operand = [&](u64 kind, u64 idx) -> e3 {
            switch (kind) {
                case 0: if (idx >= slot.size()) { ok = false; return e3_base(0); } return slot[idx];
                case 1: if (idx >= Wt) { ok = false; return e3_base(0); } return e3_make(h_ev_z[3 * idx], h_ev_z[3 * idx + 1], h_ev_z[3 * idx + 2]);
                case 2: if (idx >= Wt) { ok = false; return e3_base(0); } return e3_make(h_ev_zw[3 * idx], h_ev_zw[3 * idx + 1], h_ev_zw[3 * idx + 2]);
                case 3: if (idx >= n_fixed) { ok = false; return e3_base(0); } return fixed[idx];
                case 4: if (idx >= (size_t)n_pubchal) { ok = false; return e3_base(0); } return e3_base(h_pubchal[idx]);
                case 5: if (idx >= n_const) { ok = false; return e3_base(0); } return e3_base(consts[idx] % GL_P);
                case 6: return xml;
                default: ok = false; return e3_base(0);
            }
        };
        for (size_t i = 0; i < n_instr && ok; i++) {
            const u64 w = ins[i], op = w & 0xFF, d = (w >> 8) & 0xFFFF, ka = (w >> 24) & 0xF, ia = (w >> 28) & 0xFFFF, kb = (w >> 44) & 0xF, ib = (w >> 48) & 0xFFFF;
            const e3 a = operand(ka, ia);
            if (op == 4) {                     // OUT: constraint k_out is this value
                if (k_out >= K) return ZP_ERR_ARG;
                memcpy(h_out + 3 * k_out++, a.c, 24);
                continue;
            }
            const e3 b = operand(kb, ib);
            if (d >= slot.size()) return ZP_ERR_ARG;
            if (op == 1) slot[d] = e3_add(a, b);
            else if (op == 2) slot[d] = e3_sub(a, b);
            else if (op == 3) slot[d] = e3_mul(a, b);
            else return ZP_ERR_ARG;
        }
        return ok && k_out == K ? ZP_OK : ZP_ERR_ARG;
    } catch (...) {
        return ZP_ERR_NOMEM;
    }
}

int32_t zp_program_eval_ext(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn, uint64_t root32,
                            const uint64_t zeta3[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, int32_t n_cols, uint64_t *h_out, int32_t n_out,
                            int32_t threads) {
    return program_eval_ext_impl(h_program, program_words, h_pubchal, n_pubchal, logn, root32, zeta3, h_ev_z, h_ev_zw, n_cols, h_out, n_out, threads, nullptr, 0);
}

// The FIXED columns of a program at the out-of-domain point: h_fixed u64[n_fixed][3] (n_fixed must be the program's): columns 0 / 1 the first-row /
// last-row selectors, then the sparse periodic columns (public-input entries read from h_pubchal).  A function of (statement, public inputs, zeta)
// alone -- what the Groth16 wrap's circuit takes as committed input instead of evaluating ~10^5 entries in F_r (service/wrap_arith.py), and what a
// reader of its public input recomputes.  Same refusals as zp_program_eval_ext.
int32_t zp_program_fixed_eval_ext(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn, uint64_t root32,
                                  const uint64_t zeta3[3], uint64_t *h_fixed, int32_t n_fixed, int32_t threads) {
    if (!h_fixed) return ZP_ERR_ARG;
    return program_eval_ext_impl(h_program, program_words, h_pubchal, n_pubchal, logn, root32, zeta3, nullptr, nullptr, 0, nullptr, 0, threads, h_fixed, n_fixed);
}

}  // extern "C"
