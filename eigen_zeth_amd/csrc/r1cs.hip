// R1CS over the BN254 scalar field for the Groth16 wrap of GenFinalProof (proto/prover/v1/prover.proto:130-148; consumer
// src/settlement/ethereum/mod.rs:338-394, interfaces/zkvm.rs:82-130): witness completion, the evaluation vectors A w, B w, C w of the QAP
// step (what zp_qap_quotient_bn254 takes) and the key generator's scalars u_j(tau), v_j(tau), w_j(tau) -- HOST code: a circuit that verifies
// the hashing of a STARK is thousands of copies of ONE gadget (a width-17 Poseidon-BN254 permutation: 613 constraints over 631 local wires, about
// 12 000 matrix entries) plus a few ten thousand glue constraints, so the matrices are never materialised: the gadget is a TEMPLATE (three
// sparse matrices over local wires), an instance is a map of its 17 input wires + the base of its internal wires and constraints, the glue
// comes as explicit sparse rows.  Everything runs over the template per instance (23 M multiply-adds for 1 900 instances: a second on one
// core, instances in parallel on threads for the key scalars).  The group operations (fixed-base multiplications for the key, MSMs for a
// proof) are the GPU's (csrc/msm.hip).  Layout of the circuit blob: eigen_zeth_amd/service/r1cs.py.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "fr254.hpp"

namespace {

typedef unsigned __int128 u128;

// ---- F_r, four 64-bit limbs, Montgomery form (R = 2^256) on the host
struct Fr { uint64_t l[4]; };
const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
const Fr FR_R2 = {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}};
const Fr FR_ONE = {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}};
const uint64_t FR_N0 = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64

inline bool fr_geq_mod(const uint64_t *a) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > FR_MOD[i]) return true;
        if (a[i] < FR_MOD[i]) return false;
    }
    return true;
}
inline void fr_sub_mod(uint64_t *a) {
    u128 b = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a[i] - FR_MOD[i] - (uint64_t)b;
        a[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
inline Fr fr_add(const Fr &a, const Fr &b) {
    Fr r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (c || fr_geq_mod(r.l)) fr_sub_mod(r.l);
    return r;
}
inline Fr fr_sub(const Fr &a, const Fr &b) {
    Fr r;
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        const u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)br;
        r.l[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + FR_MOD[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
inline Fr fr_mul(const Fr &a, const Fr &b) {      // CIOS Montgomery product
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * FR_N0;
        c = (u128)m * FR_MOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * FR_MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fr_geq_mod(r.l)) fr_sub_mod(r.l);
    return r;
}
inline bool fr_is_zero(const Fr &a) { return (a.l[0] | a.l[1] | a.l[2] | a.l[3]) == 0; }
inline bool fr_eq(const Fr &a, const Fr &b) { return a.l[0] == b.l[0] && a.l[1] == b.l[1] && a.l[2] == b.l[2] && a.l[3] == b.l[3]; }
inline Fr fr_from_std(const uint64_t *w) { Fr a = {{w[0], w[1], w[2], w[3]}}; return fr_mul(a, FR_R2); }
inline void fr_to_std(const Fr &a, uint64_t *w) { const Fr one = {{1, 0, 0, 0}}; const Fr r = fr_mul(a, one); memcpy(w, r.l, 32); }
Fr fr_pow(Fr a, const uint64_t *e, int bits) {
    Fr r = FR_ONE;
    for (int i = bits - 1; i >= 0; i--) {
        r = fr_mul(r, r);
        if ((e[i >> 6] >> (i & 63)) & 1) r = fr_mul(r, a);
    }
    return r;
}
Fr fr_inv(const Fr &a) {
    uint64_t e[4] = {FR_MOD[0] - 2, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
    return fr_pow(a, e, 254);
}
inline bool std_canonical(const uint64_t *w) { return !fr_geq_mod(w); }

// ---- the circuit blob (service/r1cs.py: pack_circuit)
constexpr uint64_t MAGIC = 0x3130534331525a50ULL;   // "PZR1CS01"
constexpr uint64_t MAGIC2 = 0x3230534331525a50ULL;  // "PZR1CS02": + the arithmetic templates (service/arith.py) behind the explicit constraints
struct Mat { const uint64_t *ptr, *idx, *val; };      // CSR over constraints: ptr[n + 1], idx[nnz] (wire), val[nnz][4] (standard form)
// An arithmetic template (service/arith.py: Goldilocks arithmetic of the final STARK's verifier inside F_r, wrap stage B-2): rows over LOCAL wires
// (0 = the constant, 1..n_in inputs, then n_int internal wires) as triples of linear-combination ids into one pool, a witness program that
// DEFINES every internal wire, and n_inst instances (input wires + the base of the internal wires).  Instance i owns the constraints
// [first_row + i n_rows, first_row + (i + 1) n_rows).
constexpr uint32_t AR_UNIT = 1u << 31;               // LC id: the unit combination of local wire (id & ~AR_UNIT)
enum { AR_OP_MUL = 1, AR_OP_DIVMOD = 3, AR_OP_BITS = 4, AR_OP_INV3 = 5 };
struct ArithT {
    uint64_t n_in, n_int, n_coef, n_lc, nnz, n_rows, n_ops, n_inst, first_row;
    const uint64_t *coef, *lc_ptr, *lc_ent, *rows, *ops, *inst;
    uint64_t n_local() const { return 1 + n_in + n_int; }
    uint64_t global(const uint64_t *in, uint64_t local) const { return local == 0 ? 0 : local <= n_in ? in[local - 1] : in[n_in] + (local - 1 - n_in); }
};
struct Circ {
    std::vector<ArithT> ar;
    uint64_t n_arith_rows = 0;
    uint64_t n_wires, n_cons, logm, t, n_local, tc, n_inst, n_extra, n_pub, n_waves;
    const uint64_t *tdef, *inst, *edef, *waves;      // waves[n_waves + 1]: instances [waves[k], waves[k + 1]) read only wires set before wave k               // tdef[tc]: the local wire a template constraint defines; inst: (t + 2) words each; edef[n_extra]
    Mat T[3], E[3];
    uint64_t extra_base() const { return n_inst * tc; }   // constraints: instances first (instance i owns [i tc, (i + 1) tc)), then the extras
};

bool parse_mat(const uint64_t *d, size_t words, size_t &at, uint64_t rows, uint64_t max_idx, Mat *m) {
    if (at + rows + 1 > words) return false;
    m->ptr = d + at;
    at += rows + 1;
    const uint64_t nnz = m->ptr[rows];
    if (m->ptr[0] != 0 || nnz > (1ull << 32) || at + nnz + 4 * nnz > words) return false;
    for (uint64_t i = 0; i < rows; i++) if (m->ptr[i] > m->ptr[i + 1]) return false;
    m->idx = d + at; at += nnz;
    m->val = d + at; at += 4 * nnz;
    for (uint64_t k = 0; k < nnz; k++) if (m->idx[k] >= max_idx || !std_canonical(m->val + 4 * k)) return false;
    return true;
}

bool parse_arith(const uint64_t *d, size_t words, size_t &at, uint64_t n_tpl, Circ *c) {
    uint64_t next_row = c->n_inst * c->tc + c->n_extra;
    for (uint64_t k = 0; k < n_tpl; k++) {
        if (at + 12 > words) return false;
        ArithT t;
        t.n_in = d[at]; t.n_int = d[at + 1]; t.n_coef = d[at + 2]; t.n_lc = d[at + 3]; t.nnz = d[at + 4]; t.n_rows = d[at + 5]; t.n_ops = d[at + 6]; t.n_inst = d[at + 7];
        t.first_row = d[at + 8];
        at += 12;
        if (t.n_in > (1u << 20) || t.n_int < 1 || t.n_int > (1u << 26) || t.n_coef > (1u << 24) || t.n_lc < 1 || t.n_lc > (1u << 28) || t.nnz > (1ull << 30) || t.n_rows < 1 ||
            t.n_rows > (1u << 26) || t.n_ops > (1u << 26) || t.n_inst < 1 || t.n_inst > (1u << 16) || t.first_row != next_row)
            return false;
        const uint64_t nl = t.n_local();
        if (nl >= AR_UNIT) return false;
        const uint64_t need = 4 * t.n_coef + (t.n_lc + 1) + t.nnz + 2 * t.n_rows + 4 * t.n_ops + t.n_inst * (t.n_in + 1);
        if (at + need > words) return false;
        t.coef = d + at; at += 4 * t.n_coef;
        t.lc_ptr = d + at; at += t.n_lc + 1;
        t.lc_ent = d + at; at += t.nnz;
        t.rows = d + at; at += 2 * t.n_rows;
        t.ops = d + at; at += 4 * t.n_ops;
        t.inst = d + at; at += t.n_inst * (t.n_in + 1);
        for (uint64_t i = 0; i < t.n_coef; i++) if (!std_canonical(t.coef + 4 * i)) return false;
        if (t.lc_ptr[0] != 0 || t.lc_ptr[t.n_lc] != t.nnz || t.lc_ptr[1] != 0) return false;           // LC 0 is the empty combination
        for (uint64_t i = 0; i < t.n_lc; i++) if (t.lc_ptr[i] > t.lc_ptr[i + 1]) return false;
        for (uint64_t e = 0; e < t.nnz; e++) if ((t.lc_ent[e] & 0xFFFFFFFFull) >= nl || (t.lc_ent[e] >> 32) >= t.n_coef) return false;
        auto lc_ok = [&](uint64_t id) { return (id & AR_UNIT) ? (id & (AR_UNIT - 1)) < nl && id < (1ull << 32) : id < t.n_lc; };
        for (uint64_t q = 0; q < t.n_rows; q++)
            if (!lc_ok(t.rows[2 * q] & 0xFFFFFFFFull) || !lc_ok(t.rows[2 * q] >> 32) || !lc_ok(t.rows[2 * q + 1])) return false;
        for (uint64_t o = 0; o < t.n_ops; o++) {
            const uint64_t *op = t.ops + 4 * o;
            const uint64_t code = op[0] & 0xFF, nbits = (op[0] >> 8) & 0xFFFF, flag = (op[0] >> 24) & 0xFF, dst = op[1];
            uint64_t cnt = 1;
            if (code == AR_OP_MUL) cnt = 1;
            else if (code == AR_OP_DIVMOD) { if (flag != 0 && flag != 2) return false; cnt = flag == 2 ? 1 : 2; }
            else if (code == AR_OP_BITS) { if (nbits < 1 || nbits > 254) return false; cnt = nbits; }
            else if (code == AR_OP_INV3) cnt = 3;
            else return false;
            if (dst < 1 + t.n_in || dst + cnt > nl) return false;                                        // an op defines INTERNAL wires
            if (!lc_ok(op[2] & 0xFFFFFFFFull) || !lc_ok(op[2] >> 32) || !lc_ok(op[3])) return false;
        }
        for (uint64_t i = 0; i < t.n_inst; i++) {
            const uint64_t *in = t.inst + i * (t.n_in + 1);
            for (uint64_t j = 0; j < t.n_in; j++) if (in[j] >= c->n_wires) return false;
            if (in[t.n_in] + t.n_int > c->n_wires || in[t.n_in] < 1 + c->n_pub) return false;
        }
        next_row += t.n_inst * t.n_rows;
        c->n_arith_rows += t.n_inst * t.n_rows;
        c->ar.push_back(t);
    }
    return true;
}

bool parse(const uint64_t *d, size_t words, Circ *c) {
    if (!d || words < 16 || (d[0] != MAGIC && d[0] != MAGIC2)) return false;
    const uint64_t n_tpl = d[0] == MAGIC2 ? d[11] : 0;
    if (n_tpl > 16) return false;
    c->ar.clear();
    c->n_arith_rows = 0;
    c->n_wires = d[1]; c->n_cons = d[2]; c->logm = d[3]; c->t = d[4]; c->n_local = d[5]; c->tc = d[6]; c->n_inst = d[7]; c->n_extra = d[8]; c->n_pub = d[9];
    c->n_waves = d[10];
    if (c->n_wires < 2 || c->n_wires > (1ull << 28) || c->logm > 28 || c->t < 2 || c->t > 64 || c->n_local < 1 + c->t || c->n_local > (1u << 20) || c->tc < 1 ||
        c->tc > (1u << 20) || c->n_inst > (1u << 24) || c->n_extra > (1ull << 28) || c->n_pub < 1 || 1 + c->n_pub > c->n_wires)
        return false;
    if (c->n_cons < c->n_inst * c->tc + c->n_extra || c->n_cons > (1ull << c->logm)) return false;
    size_t at = 16;
    if (at + c->tc > words) return false;
    c->tdef = d + at; at += c->tc;
    for (uint64_t i = 0; i < c->tc; i++) if (c->tdef[i] != ~0ull && (c->tdef[i] < 1 + c->t || c->tdef[i] >= c->n_local)) return false;
    for (int k = 0; k < 3; k++) if (!parse_mat(d, words, at, c->tc, c->n_local, &c->T[k])) return false;
    if (at + c->n_inst * (c->t + 2) > words) return false;
    c->inst = d + at; at += c->n_inst * (c->t + 2);
    const uint64_t n_int = c->n_local - 1 - c->t;
    for (uint64_t i = 0; i < c->n_inst; i++) {
        const uint64_t *in = c->inst + i * (c->t + 2);
        for (uint64_t k = 0; k < c->t; k++) if (in[k] >= c->n_wires) return false;
        if (in[c->t] + n_int > c->n_wires || in[c->t + 1] != i * c->tc) return false;
    }
    if (c->n_waves < 1 || c->n_waves > c->n_inst + 1 || at + c->n_waves + 1 > words) return false;
    c->waves = d + at; at += c->n_waves + 1;
    if (c->waves[0] != 0 || c->waves[c->n_waves] != c->n_inst) return false;
    for (uint64_t k = 0; k < c->n_waves; k++) if (c->waves[k] > c->waves[k + 1]) return false;
    if (at + c->n_extra > words) return false;
    c->edef = d + at; at += c->n_extra;
    for (uint64_t q = 0; q < c->n_extra; q++) if (c->edef[q] != ~0ull && c->edef[q] >= c->n_wires) return false;
    for (int k = 0; k < 3; k++) if (!parse_mat(d, words, at, c->n_extra, c->n_wires, &c->E[k])) return false;
    if (!parse_arith(d, words, at, n_tpl, c)) return false;
    return at == words && c->n_cons == c->n_inst * c->tc + c->n_extra + c->n_arith_rows;
}

// ---- the witness programs of the arithmetic templates, on the host.  W u64[n_wires][4]: wire values in standard form; set[n_wires].
// Goldilocks on the host for the one op that needs it (the inverse in F_p^3)
constexpr uint64_t GLP = 0xFFFFFFFF00000001ULL;
inline uint64_t glh_mul(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % GLP); }
inline uint64_t glh_add(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % GLP); }
inline uint64_t glh_sub(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + GLP - b) % GLP); }
uint64_t glh_inv(uint64_t a) {
    uint64_t r = 1, e = GLP - 2;
    while (e) { if (e & 1) r = glh_mul(r, a); a = glh_mul(a, a); e >>= 1; }
    return r;
}
// y with x y = 1 in F_p[t] / (t^3 - t - 1): Cramer's rule on the multiplication matrix of x (columns: x, x t, x t^2 in the basis 1, t, t^2)
bool glh_e3_inv(const uint64_t x[3], uint64_t y[3]) {
    const uint64_t a0 = x[0], a1 = x[1], a2 = x[2];
    const uint64_t m[3][3] = {{a0, a2, a1}, {a1, glh_add(a0, a2), glh_add(a1, a2)}, {a2, a1, glh_add(a0, a2)}};
    auto det2 = [&](int r0, int r1, int c0, int c1) { return glh_sub(glh_mul(m[r0][c0], m[r1][c1]), glh_mul(m[r0][c1], m[r1][c0])); };
    const uint64_t c00 = det2(1, 2, 1, 2), c01 = det2(1, 2, 0, 2), c02 = det2(1, 2, 0, 1);
    const uint64_t det = glh_add(glh_sub(glh_mul(m[0][0], c00), glh_mul(m[0][1], c01)), glh_mul(m[0][2], c02));
    if (det == 0) return false;
    const uint64_t di = glh_inv(det);
    // the first column of the inverse matrix = the solution of M y = e_0: cofactors of the first ROW, signs alternating
    y[0] = glh_mul(c00, di);
    y[1] = glh_mul(glh_sub(0, c01), di);
    y[2] = glh_mul(c02, di);
    return true;
}
struct ArithCoefs { std::vector<Fr> m; };       // a template's coefficients in Montgomery form: fr_mul(coef, standard-form wire) = the standard-form product
void arith_coefs(const ArithT &t, ArithCoefs *out) {
    out->m.resize(t.n_coef);
    for (uint64_t i = 0; i < t.n_coef; i++) out->m[i] = fr_from_std(t.coef + 4 * i);
}
// value of LC `id` of instance `in` over W (standard form); false: it reads a wire nobody has set
inline bool arith_lc(const ArithT &t, const ArithCoefs &cf, const uint64_t *in, uint32_t id, const uint64_t *W, const uint8_t *set, Fr *out) {
    if (id & AR_UNIT) {
        const uint64_t g = t.global(in, id & (AR_UNIT - 1));
        if (!set[g]) return false;
        memcpy(out->l, W + 4 * g, 32);
        return true;
    }
    Fr acc = {{0, 0, 0, 0}};
    for (uint64_t e = t.lc_ptr[id]; e < t.lc_ptr[id + 1]; e++) {
        const uint64_t g = t.global(in, t.lc_ent[e] & 0xFFFFFFFFull);
        if (!set[g]) return false;
        Fr w;
        memcpy(w.l, W + 4 * g, 32);
        acc = fr_add(acc, fr_mul(cf.m[t.lc_ent[e] >> 32], w));
    }
    *out = acc;
    return true;
}
// runs the witness program of instance i: 0 = done, -20 = an op cannot be carried out (no witness: the statement is false), -21 = unset input
int32_t arith_witness(const ArithT &t, const ArithCoefs &cf, uint64_t i, uint64_t *W, uint8_t *set) {
    const uint64_t *in = t.inst + i * (t.n_in + 1);
    auto put = [&](uint64_t local, const uint64_t *v4) {
        const uint64_t g = t.global(in, local);
        memcpy(W + 4 * g, v4, 32);
        set[g] = 1;
    };
    for (uint64_t o = 0; o < t.n_ops; o++) {
        const uint64_t *op = t.ops + 4 * o;
        const uint64_t code = op[0] & 0xFF, nbits = (op[0] >> 8) & 0xFFFF, flag = (op[0] >> 24) & 0xFF, dst = op[1];
        Fr a, b, c3;
        if (!arith_lc(t, cf, in, (uint32_t)(op[2] & 0xFFFFFFFFull), W, set, &a)) return -21;
        if (code == AR_OP_MUL) {
            if (!arith_lc(t, cf, in, (uint32_t)(op[2] >> 32), W, set, &b)) return -21;
            const Fr pr = fr_mul(fr_mul(a, FR_R2), b);
            put(dst, pr.l);
        } else if (code == AR_OP_DIVMOD) {
            uint64_t q[4], rem = 0;
            for (int k = 3; k >= 0; k--) {
                const u128 cur = ((u128)rem << 64) | a.l[k];
                q[k] = (uint64_t)(cur / GLP);
                rem = (uint64_t)(cur % GLP);
            }
            if (flag == 2) {
                if (rem) return -20;
                put(dst, q);
            } else {
                const uint64_t r4[4] = {rem, 0, 0, 0};
                put(dst, q);
                put(dst + 1, r4);
            }
        } else if (code == AR_OP_BITS) {
            for (uint64_t k = nbits; k < 256; k++)
                if ((a.l[k >> 6] >> (k & 63)) & 1) return -20;
            for (uint64_t k = 0; k < nbits; k++) {
                const uint64_t v4[4] = {(a.l[k >> 6] >> (k & 63)) & 1, 0, 0, 0};
                put(dst + k, v4);
            }
        } else {        // AR_OP_INV3
            if (!arith_lc(t, cf, in, (uint32_t)(op[2] >> 32), W, set, &b) || !arith_lc(t, cf, in, (uint32_t)op[3], W, set, &c3)) return -21;
            auto modp = [](const Fr &v) {
                uint64_t rem = 0;
                for (int k = 3; k >= 0; k--) rem = (uint64_t)((((u128)rem << 64) | v.l[k]) % GLP);
                return rem;
            };
            const uint64_t x[3] = {modp(a), modp(b), modp(c3)};
            uint64_t y[3];
            if (!glh_e3_inv(x, y)) return -20;
            for (int k = 0; k < 3; k++) {
                const uint64_t v4[4] = {y[k], 0, 0, 0};
                put(dst + k, v4);
            }
        }
    }
    return ZP_OK;
}
// every instance of every template whose internal wires the caller did not set, in blob order (a template's instances on threads: they read
// caller-set wires and wires of EARLIER templates, and write their own).  *bad: the first row of the instance that has no witness.
int32_t arith_witness_all(const Circ &c, uint64_t *W, uint8_t *set, int64_t *bad) {
    for (const ArithT &t : c.ar) {
        ArithCoefs cf;
        arith_coefs(t, &cf);
        std::atomic<int64_t> fail{-1};
        std::atomic<int32_t> code{ZP_OK};
        std::atomic<uint64_t> next{0};
        auto body = [&]() {
            for (;;) {
                const uint64_t i = next.fetch_add(1);
                if (i >= t.n_inst) return;
                if (set[t.inst[i * (t.n_in + 1) + t.n_in]]) continue;           // the caller brought this instance's wires (a complete witness)
                const int32_t rc = arith_witness(t, cf, i, W, set);
                if (rc != ZP_OK) {
                    int64_t cur = fail.load();
                    const int64_t row = (int64_t)(t.first_row + i * t.n_rows);
                    while ((cur < 0 || row < cur) && !fail.compare_exchange_weak(cur, row)) {}
                    code.store(rc);
                }
            }
        };
        int T = (int)std::thread::hardware_concurrency();
        if (T > 16) T = 16;
        if ((uint64_t)T > t.n_inst) T = (int)t.n_inst;
        if (T <= 1) body();
        else {
            std::vector<std::thread> pool;
            for (int k = 0; k < T; k++) pool.emplace_back(body);
            for (auto &th : pool) th.join();
        }
        if (fail.load() >= 0) { if (bad) *bad = fail.load(); return code.load(); }
    }
    return ZP_OK;
}

inline uint64_t local_to_global(const Circ &c, const uint64_t *in, uint64_t local) {
    if (local == 0) return 0;
    if (local <= c.t) return in[local - 1];
    return in[c.t] + (local - 1 - c.t);
}

}  // namespace

extern "C" {

// Completes and checks a witness.  witness u64[n_wires][4] (standard form): wire 0 = 1, the public inputs and every wire that is not internal
// to a gadget instance are set by the caller; set[n_wires] (bytes) says which.  The internal wires of every instance are computed in instance
// order (a template constraint defines one: wire = (A w)(B w) - (rest of C w), C's coefficient of it being 1); a wire that is defined twice
// must agree.  Then EVERY constraint is checked.  a_ev / b_ev / c_ev u64[2^logm][4] receive A w, B w, C w per constraint (zero padded): the
// input of the QAP step.  Returns ZP_OK; -20 = the assignment does not satisfy the circuit (*bad = first violated constraint) -- there is no
// proof for a false statement; -21 = an instance reads a wire nobody has set; ZP_ERR_ARG = malformed blob.
int32_t zp_r1cs_eval(const uint64_t *circ, size_t words, uint64_t *witness, uint8_t *set, uint64_t *a_ev, uint64_t *b_ev, uint64_t *c_ev, int64_t *bad) {
    Circ c;
    if (!parse(circ, words, &c) || !witness || !set || !a_ev || !b_ev || !c_ev) return ZP_ERR_ARG;
    if (bad) *bad = -1;
    try {
        if (!c.ar.empty()) {
            for (uint64_t j = 0; j < c.n_wires; j++) if (set[j] && !std_canonical(witness + 4 * j)) return ZP_ERR_ARG;
            const int32_t arc = arith_witness_all(c, witness, set, bad);
            if (arc != ZP_OK) return arc;
        }
        std::vector<Fr> w(c.n_wires);
        for (uint64_t j = 0; j < c.n_wires; j++)
            if (set[j]) {
                if (!std_canonical(witness + 4 * j)) return ZP_ERR_ARG;
                w[j] = fr_from_std(witness + 4 * j);
            }
        if (!set[0] || !fr_eq(w[0], FR_ONE)) return ZP_ERR_ARG;
        const uint64_t m = 1ull << c.logm;
        memset(a_ev, 0, m * 32); memset(b_ev, 0, m * 32); memset(c_ev, 0, m * 32);
        std::vector<Fr> tv[3];                 // template coefficients in Montgomery form, once
        for (int k = 0; k < 3; k++) {
            const uint64_t nnz = c.T[k].ptr[c.tc];
            tv[k].resize(nnz);
            for (uint64_t e = 0; e < nnz; e++) tv[k][e] = fr_from_std(c.T[k].val + 4 * e);
        }
        std::atomic<int64_t> first_bad{-1}, unset{-1};
        auto note = [](std::atomic<int64_t> &slot, int64_t v) {          // keep the smallest index
            int64_t cur = slot.load();
            while ((cur < 0 || v < cur) && !slot.compare_exchange_weak(cur, v)) {}
        };
        auto run_instance = [&](uint64_t i) {
            const uint64_t *in = c.inst + i * (c.t + 2);
            for (uint64_t k = 0; k < c.t; k++) if (!set[in[k]]) { note(unset, (int64_t)(i * c.tc)); return; }
            for (uint64_t q = 0; q < c.tc; q++) {
                Fr s[3];
                const uint64_t def = c.tdef[q];
                const uint64_t gdef = def == ~0ull ? ~0ull : local_to_global(c, in, def);
                for (int k = 0; k < 3; k++) {
                    Fr acc = {{0, 0, 0, 0}};
                    for (uint64_t e = c.T[k].ptr[q]; e < c.T[k].ptr[q + 1]; e++) {
                        const uint64_t g = local_to_global(c, in, c.T[k].idx[e]);
                        if (k == 2 && g == gdef && !set[g]) continue;       // the wire this constraint defines (coefficient 1 in C)
                        if (!set[g]) { note(unset, (int64_t)(i * c.tc + q)); return; }
                        acc = fr_add(acc, fr_mul(tv[k][e], w[g]));
                    }
                    s[k] = acc;
                }
                const Fr ab = fr_mul(s[0], s[1]);
                if (gdef != ~0ull && !set[gdef]) {
                    w[gdef] = fr_sub(ab, s[2]);
                    set[gdef] = 1;
                    s[2] = ab;
                } else if (!fr_eq(ab, s[2])) {
                    note(first_bad, (int64_t)(i * c.tc + q));
                }
                const uint64_t row = i * c.tc + q;
                fr_to_std(s[0], a_ev + 4 * row); fr_to_std(s[1], b_ev + 4 * row); fr_to_std(s[2], c_ev + 4 * row);
            }
        };
        // instances of one wave read only wires set before it and define disjoint (their own internal) wires: threads
        int T = (int)std::thread::hardware_concurrency();
        if (T < 1) T = 1;
        if (T > 32) T = 32;
        for (uint64_t wv = 0; wv < c.n_waves; wv++) {
            const uint64_t i0 = c.waves[wv], i1 = c.waves[wv + 1];
            if (i1 - i0 < 4 || T == 1) {
                for (uint64_t i = i0; i < i1; i++) run_instance(i);
            } else {
                std::atomic<uint64_t> next{i0};
                auto body = [&]() { for (;;) { const uint64_t i = next.fetch_add(1); if (i >= i1) return; run_instance(i); } };
                std::vector<std::thread> pool;
                const int nt = (uint64_t)T < i1 - i0 ? T : (int)(i1 - i0);
                for (int k = 0; k < nt; k++) pool.emplace_back(body);
                for (auto &th : pool) th.join();
            }
            if (unset.load() >= 0) { if (bad) *bad = unset.load(); return -21; }
        }
        for (uint64_t q = 0; q < c.n_extra; q++) {          // in order: an extra constraint may define a wire (C's coefficient of it being 1)
            Fr s[3];
            const uint64_t gdef = c.edef[q];
            for (int k = 0; k < 3; k++) {
                Fr acc = {{0, 0, 0, 0}};
                for (uint64_t e = c.E[k].ptr[q]; e < c.E[k].ptr[q + 1]; e++) {
                    const uint64_t g = c.E[k].idx[e];
                    if (k == 2 && g == gdef && !set[g]) continue;
                    if (!set[g]) { if (bad) *bad = (int64_t)(c.extra_base() + q); return -21; }
                    acc = fr_add(acc, fr_mul(fr_from_std(c.E[k].val + 4 * e), w[g]));
                }
                s[k] = acc;
            }
            const Fr ab = fr_mul(s[0], s[1]);
            if (gdef != ~0ull && !set[gdef]) {
                w[gdef] = fr_sub(ab, s[2]);
                set[gdef] = 1;
                s[2] = ab;
            } else if (!fr_eq(ab, s[2])) {
                note(first_bad, (int64_t)(c.extra_base() + q));
            }
            const uint64_t row = c.extra_base() + q;
            fr_to_std(s[0], a_ev + 4 * row); fr_to_std(s[1], b_ev + 4 * row); fr_to_std(s[2], c_ev + 4 * row);
        }
        for (const ArithT &t : c.ar) {          // the arithmetic templates' rows (their wires are all set: the witness programs ran first)
            std::vector<Fr> cm(t.n_coef);
            for (uint64_t i = 0; i < t.n_coef; i++) cm[i] = fr_from_std(t.coef + 4 * i);
            std::atomic<int64_t> unset_row{-1};
            auto rows_of = [&](uint64_t i) {
                const uint64_t *in = t.inst + i * (t.n_in + 1);
                for (uint64_t q = 0; q < t.n_rows; q++) {
                    const uint32_t ids[3] = {(uint32_t)(t.rows[2 * q] & 0xFFFFFFFFull), (uint32_t)(t.rows[2 * q] >> 32), (uint32_t)t.rows[2 * q + 1]};
                    Fr sv[3];
                    bool ok = true;
                    for (int k = 0; k < 3 && ok; k++) {
                        if (ids[k] & AR_UNIT) {
                            const uint64_t g = t.global(in, ids[k] & (AR_UNIT - 1));
                            ok = set[g] != 0;
                            sv[k] = w[g];
                        } else {
                            Fr acc = {{0, 0, 0, 0}};
                            for (uint64_t e = t.lc_ptr[ids[k]]; e < t.lc_ptr[ids[k] + 1] && ok; e++) {
                                const uint64_t g = t.global(in, t.lc_ent[e] & 0xFFFFFFFFull);
                                ok = set[g] != 0;
                                acc = fr_add(acc, fr_mul(cm[t.lc_ent[e] >> 32], w[g]));
                            }
                            sv[k] = acc;
                        }
                    }
                    const uint64_t row = t.first_row + i * t.n_rows + q;
                    if (!ok) { note(unset_row, (int64_t)row); return; }
                    if (!fr_eq(fr_mul(sv[0], sv[1]), sv[2])) note(first_bad, (int64_t)row);
                    fr_to_std(sv[0], a_ev + 4 * row); fr_to_std(sv[1], b_ev + 4 * row); fr_to_std(sv[2], c_ev + 4 * row);
                }
            };
            std::atomic<uint64_t> next{0};
            auto body = [&]() { for (;;) { const uint64_t i = next.fetch_add(1); if (i >= t.n_inst) return; rows_of(i); } };
            const int nt = (uint64_t)T < t.n_inst ? T : (int)t.n_inst;
            if (nt <= 1) body();
            else {
                std::vector<std::thread> pool;
                for (int k = 0; k < nt; k++) pool.emplace_back(body);
                for (auto &th : pool) th.join();
            }
            if (unset_row.load() >= 0) { if (bad) *bad = unset_row.load(); return -21; }
        }
        for (uint64_t j = 0; j < c.n_wires; j++) {
            if (!set[j]) { if (bad) *bad = (int64_t)j; return -21; }
            fr_to_std(w[j], witness + 4 * j);
        }
        if (first_bad.load() >= 0) { if (bad) *bad = first_bad.load(); return -20; }
        return ZP_OK;
    } catch (...) {
        return ZP_ERR_NOMEM;
    }
}

// The scalars of a Groth16 key for this circuit at the point tau (a LOCAL, seeded setup: toxic waste known -- test keys, not a ceremony):
// with L_i the Lagrange basis of the 2^logm-th roots of unity (root 5^((r-1)/2^logm), the snarkjs / circom convention),
//   u_j = sum_i A[i][j] L_i(tau), v_j = sum_i B[i][j] L_i(tau), w_j = sum_i C[i][j] L_i(tau)            out_u, out_v u64[n_wires][4]
//   l_j = (beta u_j + alpha v_j + w_j) / delta for private wires, / gamma for wire 0 and the public inputs   out_l u64[n_wires][4]
//   h_i = tau^i (tau^m - 1) / delta, i < m - 1                                                              out_h u64[2^logm - 1][4]
// params u64[5][4]: tau, alpha, beta, gamma, delta (standard form).  The group elements are these scalars times the generators:
// zp_fixed_base_mul_bn254 / _g2.
int32_t zp_r1cs_key_scalars(const uint64_t *circ, size_t words, const uint64_t *params, uint64_t *out_u, uint64_t *out_v, uint64_t *out_l, uint64_t *out_h,
                            int32_t threads) {
    Circ c;
    if (!parse(circ, words, &c) || !params || !out_u || !out_v || !out_l || !out_h) return ZP_ERR_ARG;
    for (int k = 0; k < 5; k++) if (!std_canonical(params + 4 * k)) return ZP_ERR_ARG;
    try {
        const Fr tau = fr_from_std(params), alpha = fr_from_std(params + 4), beta = fr_from_std(params + 8), gamma = fr_from_std(params + 12),
                 delta = fr_from_std(params + 16);
        if (fr_is_zero(gamma) || fr_is_zero(delta)) return ZP_ERR_ARG;
        const uint64_t m = 1ull << c.logm;
        // omega = 5^((r - 1) / 2^logm)
        uint64_t e[4] = {FR_MOD[0] - 1, FR_MOD[1], FR_MOD[2], FR_MOD[3]};
        for (uint64_t s = 0; s < c.logm; s++) {                     // (r - 1) >> logm
            for (int i = 0; i < 4; i++) e[i] = (e[i] >> 1) | (i < 3 ? e[i + 1] << 63 : 0);
        }
        const uint64_t five[4] = {5, 0, 0, 0};
        const Fr omega = fr_pow(fr_from_std(five), e, 254);
        // L_i(tau) = (tau^m - 1) w^i / (m (tau - w^i)): batch inversion of (tau - w^i)
        std::vector<Fr> L(m), pre(m);
        Fr wi = FR_ONE, run = FR_ONE;
        for (uint64_t i = 0; i < m; i++) {
            L[i] = fr_sub(tau, wi);
            if (fr_is_zero(L[i])) return ZP_ERR_ARG;               // tau on the domain: not a usable point
            pre[i] = run;
            run = fr_mul(run, L[i]);
            wi = fr_mul(wi, omega);
        }
        Fr tm = tau;
        for (uint64_t s = 0; s < c.logm; s++) tm = fr_mul(tm, tm);
        const Fr zt = fr_sub(tm, FR_ONE);
        const uint64_t mm[4] = {m, 0, 0, 0};
        const Fr scale = fr_mul(zt, fr_inv(fr_from_std(mm)));
        Fr inv = fr_inv(run);
        {
            std::vector<Fr> wpow(m);
            Fr x = FR_ONE;
            for (uint64_t i = 0; i < m; i++) { wpow[i] = x; x = fr_mul(x, omega); }
            for (uint64_t i = m; i-- > 0;) {
                const Fr d = fr_mul(inv, pre[i]);                   // 1 / (tau - w^i)
                inv = fr_mul(inv, L[i]);
                L[i] = fr_mul(fr_mul(scale, wpow[i]), d);
            }
        }
        std::vector<Fr> tv[3];
        for (int k = 0; k < 3; k++) {
            const uint64_t nnz = c.T[k].ptr[c.tc];
            tv[k].resize(nnz);
            for (uint64_t q = 0; q < nnz; q++) tv[k][q] = fr_from_std(c.T[k].val + 4 * q);
        }
        const Fr zero = {{0, 0, 0, 0}};
        std::vector<Fr> acc[3];
        for (int k = 0; k < 3; k++) acc[k].assign(c.n_wires, zero);
        // instances own their internal wires (no two write the same accumulator) but share input wires and the constant: a thread adds the
        // internal ones in place and keeps what it has for shared wires in a list of its own, merged afterwards
        int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
        if (T < 1) T = 1;
        if (T > 32) T = 32;
        if ((uint64_t)T > c.n_inst) T = (int)(c.n_inst ? c.n_inst : 1);
        struct Shared { uint64_t g; int k; Fr v; };
        std::vector<std::vector<Shared>> side(T);
        auto work = [&](int tid, uint64_t i0, uint64_t i1) {
            for (uint64_t i = i0; i < i1; i++) {
                const uint64_t *in = c.inst + i * (c.t + 2);
                for (int k = 0; k < 3; k++)
                    for (uint64_t q = 0; q < c.tc; q++) {
                        const Fr &Li = L[i * c.tc + q];
                        for (uint64_t x = c.T[k].ptr[q]; x < c.T[k].ptr[q + 1]; x++) {
                            const uint64_t lw = c.T[k].idx[x];
                            const Fr v = fr_mul(tv[k][x], Li);
                            if (lw <= c.t) side[tid].push_back(Shared{local_to_global(c, in, lw), k, v});
                            else { Fr &a = acc[k][in[c.t] + (lw - 1 - c.t)]; a = fr_add(a, v); }
                        }
                    }
            }
        };
        {
            std::vector<std::thread> pool;
            for (int tid = 0; tid < T; tid++) {
                const uint64_t i0 = c.n_inst * tid / T, i1 = c.n_inst * (tid + 1) / T;
                if (i0 < i1) pool.emplace_back(work, tid, i0, i1);
            }
            for (auto &th : pool) th.join();
        }
        for (const auto &lst : side)
            for (const Shared &e : lst) acc[e.k][e.g] = fr_add(acc[e.k][e.g], e.v);
        for (int k = 0; k < 3; k++)
            for (uint64_t q = 0; q < c.n_extra; q++) {
                const Fr &Li = L[c.extra_base() + q];
                for (uint64_t x = c.E[k].ptr[q]; x < c.E[k].ptr[q + 1]; x++)
                    acc[k][c.E[k].idx[x]] = fr_add(acc[k][c.E[k].idx[x]], fr_mul(fr_from_std(c.E[k].val + 4 * x), Li));
            }
        for (const ArithT &t : c.ar) {          // rows of the arithmetic templates: instance by instance over the template's combinations
            std::vector<Fr> cm(t.n_coef);
            for (uint64_t i = 0; i < t.n_coef; i++) cm[i] = fr_from_std(t.coef + 4 * i);
            for (uint64_t i = 0; i < t.n_inst; i++) {
                const uint64_t *in = t.inst + i * (t.n_in + 1);
                for (uint64_t q = 0; q < t.n_rows; q++) {
                    const Fr &Li = L[t.first_row + i * t.n_rows + q];
                    const uint32_t ids[3] = {(uint32_t)(t.rows[2 * q] & 0xFFFFFFFFull), (uint32_t)(t.rows[2 * q] >> 32), (uint32_t)t.rows[2 * q + 1]};
                    for (int k = 0; k < 3; k++) {
                        if (ids[k] & AR_UNIT) {
                            Fr &a = acc[k][t.global(in, ids[k] & (AR_UNIT - 1))];
                            a = fr_add(a, Li);
                        } else {
                            for (uint64_t e = t.lc_ptr[ids[k]]; e < t.lc_ptr[ids[k] + 1]; e++) {
                                Fr &a = acc[k][t.global(in, t.lc_ent[e] & 0xFFFFFFFFull)];
                                a = fr_add(a, fr_mul(cm[t.lc_ent[e] >> 32], Li));
                            }
                        }
                    }
                }
            }
        }
        const Fr dinv = fr_inv(delta), ginv = fr_inv(gamma);
        for (uint64_t j = 0; j < c.n_wires; j++) {
            fr_to_std(acc[0][j], out_u + 4 * j);
            fr_to_std(acc[1][j], out_v + 4 * j);
            const Fr l = fr_add(fr_add(fr_mul(beta, acc[0][j]), fr_mul(alpha, acc[1][j])), acc[2][j]);
            fr_to_std(fr_mul(l, j <= c.n_pub ? ginv : dinv), out_l + 4 * j);
        }
        Fr tp = fr_mul(zt, dinv);
        for (uint64_t i = 0; i + 1 < m; i++) { fr_to_std(tp, out_h + 4 * i); tp = fr_mul(tp, tau); }
        return ZP_OK;
    } catch (...) {
        return ZP_ERR_NOMEM;
    }
}

}  // extern "C"

// ======================================================================================================================
// The Groth16 wrap behind two calls (GenFinalProof, proto/prover/v1/prover.proto:130-148; src/prover/provider.rs:472-503): what
// eigen_zeth_amd/service/groth16.py orchestrated in Python through round 3.
//   zp_wrap_assign   : the caller-set wires of the wrap circuit from the binary openings of the final STARK (zp_stark_openings), driven by the
//                      assignment script the circuit builder writes beside the circuit blob (service/wrap_circuit.py: WrapCircuit.script)
//   zp_groth16_prove : witness completion + A w, B w, C w (zp_r1cs_eval), the QAP quotient on the GPU, five MSMs over the key's points in HBM,
//                      the blinding terms -> the three proof elements
namespace {

constexpr uint64_t SCRIPT_MAGIC = 0x3353504152575a50ULL;   // "PZWRAPS3" (round 6: challenge elements, a list of aux elements)
constexpr uint64_t OPEN_MAGIC = 0x33304e45504f5a50ULL;     // "PZOPEN03" (round 6: + the rate element behind every challenge)

struct OpenTree { uint64_t width, leaves, levels; const uint64_t *root; size_t off; };   // off: word offset of this tree's part inside a query record
struct Openings {
    uint64_t nq, ntr, logm;
    std::vector<OpenTree> tr;
    const uint64_t *q0;
    size_t qwords;
    // the transcript of the proof (round 5): every block of 16 field elements its sponge absorbed, in order, then the 16 rate elements of the
    // last absorbing permutation and of the squeeze-only permutations behind it (csrc/prove.hip writes them, service/wrap_circuit.py
    // TranscriptLog states them); 4 words per element
    uint64_t n_blocks, n_rates, n_chal;
    const uint64_t *blocks, *rates, *caps;      // caps: the capacity after every permutation (n_blocks + n_rates - 1 of them)
    const uint64_t *chal;                        // rate element 1 after every absorbed segment: what the challenge squeezed there is read from
    const uint64_t *query(uint64_t q) const { return q0 + q * qwords; }
};
bool parse_openings(const uint64_t *d, size_t words, Openings *o) {
    if (!d || words < 4 || d[0] != OPEN_MAGIC) return false;
    o->nq = d[1]; o->ntr = d[2]; o->logm = d[3];
    if (o->nq < 1 || o->nq > 4096 || o->ntr < 2 || o->ntr > 64 || o->logm > 40 || words < 4 + 7 * o->ntr) return false;
    size_t off = 1;
    o->tr.resize(o->ntr);
    for (uint64_t t = 0; t < o->ntr; t++) {
        OpenTree &T = o->tr[t];
        T.width = d[4 + 3 * t]; T.leaves = d[5 + 3 * t]; T.levels = d[6 + 3 * t];
        if (T.width < 1 || T.width > (1u << 16) || T.leaves < 1 || T.leaves > (1ull << 40) || (T.leaves & (T.leaves - 1)) || T.levels > 12) return false;
        T.root = d + 4 + 3 * o->ntr + 4 * t;
        T.off = off;
        off += T.width + T.levels * 64;
    }
    o->qwords = off;
    o->q0 = d + 4 + 7 * o->ntr;
    const size_t at = 4 + 7 * o->ntr + o->nq * off;
    if (words < at + 2) return false;
    o->n_blocks = d[at]; o->n_rates = d[at + 1];
    if (o->n_blocks < 1 || o->n_blocks > 4096 || o->n_rates < 1 || o->n_rates > 256) return false;
    o->blocks = d + at + 2;
    o->rates = o->blocks + o->n_blocks * 64;
    o->caps = o->rates + o->n_rates * 64;
    const size_t at2 = at + 2 + (o->n_blocks + o->n_rates) * 64 + (o->n_blocks + o->n_rates - 1) * 4;
    if (words < at2 + 1) return false;
    o->n_chal = d[at2];
    if (o->n_chal > 256) return false;
    o->chal = d + at2 + 1;
    return words == at2 + 1 + o->n_chal * 4;
}
// the scalar field's modulus r, little-endian words, and bit i of a 4-word value
constexpr uint64_t FR_R[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
inline uint64_t bit_of(const uint64_t *w4, unsigned i) { return (w4[i >> 6] >> (i & 63)) & 1; }

// one sponge block of a leaf: 56 Goldilocks values in 16 field elements (csrc/poseidon_bn254.hip: leaf_block_element)
void pack_element(const uint64_t *vals, uint64_t width, uint64_t block, int e, uint64_t *w4) {
    const uint64_t base = 56 * block;
    for (int c = 0; c < 3; c++) w4[c] = base + 3 * e + c < width ? vals[base + 3 * e + c] : 0;
    const uint64_t x = base + 48 + (e >> 1);
    w4[3] = 0;
    if (x < width) w4[3] = (e & 1) ? (vals[x] >> 32) : (vals[x] & 0xFFFFFFFFull);
}


#ifndef ZP_R1CS_HOST_ONLY      /* (tests/test_r1cs_fuzz.py builds the host half of this file with g++ under ASan + UBSan) */
// ---- witness completion and A w, B w, C w ON THE GPU (zp_r1cs_eval_device): the caller-set wires are scattered into the witness in HBM, the
// instances of the gadget are evaluated wave by wave by the permutation kernel of csrc/poseidon_bn254.hip (an instance's internal wires and
// rows ARE the intermediate values of its permutation), the explicit constraints by a sparse-row kernel.  Nothing but the few thousand set
// wires crosses PCIe, and the witness is where the MSMs read their scalars.
__global__ void __launch_bounds__(256) r1cs_scatter_kernel(const u64 *__restrict__ idx, const u64 *__restrict__ val, size_t n, u64 *__restrict__ w,
                                                           unsigned char *__restrict__ set) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 j = idx[i];
#pragma unroll
    for (int k = 0; k < 4; k++) w[j * 4 + k] = val[i * 4 + k];
    set[j] = 1;
}
// out[k] = w[wires[k]]: the scalars of an MSM over the key points of a SUBSET of the wires (those with a non-zero column in B)
__global__ void __launch_bounds__(256) r1cs_gather_kernel(const u64 *__restrict__ w, const u32 *__restrict__ wires, size_t n, u64 *__restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u64 j = wires[i];
#pragma unroll
    for (int k = 0; k < 4; k++) out[i * 4 + k] = w[j * 4 + k];
}
struct DevMat { const u64 *ptr, *idx, *val; };
// sum of one sparse row over the witness, standard form in and out (coefficient to Montgomery form, times the standard-form wire = standard form)
__device__ fr r1cs_row(const DevMat &m, u64 q, const u64 *w, const unsigned char *set, u64 skip, bool *unset) {
    fr acc = fr_zero();
    for (u64 e = m.ptr[q]; e < m.ptr[q + 1]; e++) {
        const u64 g = m.idx[e];
        if (g == skip) continue;
        if (!set[g]) *unset = true;
        acc = fr_add(acc, fr_mul(fr_to_mont(fr_from_u64(m.val + 4 * e)), fr_from_u64(w + 4 * g)));
    }
    return acc;
}
__device__ void r1cs_store(u64 *dst, u64 index, const fr &x) {
    u64 v[4];
    fr_to_u64(x, v);
#pragma unroll
    for (int k = 0; k < 4; k++) dst[index * 4 + k] = v[k];
}
// the extra constraints that DEFINE a wire nobody set, in order, by one lane (a handful per circuit: the public input of the wrap)
__global__ void r1cs_define_kernel(DevMat A, DevMat B, DevMat C, const u64 *__restrict__ defs, size_t n_defs, const u64 *__restrict__ edef, u64 row_base, u64 *w,
                                   unsigned char *set, unsigned long long *flags) {
    if (blockIdx.x || threadIdx.x) return;
    for (size_t k = 0; k < n_defs; k++) {
        const u64 q = defs[k], g = edef[q];
        if (set[g]) continue;
        bool unset = false;
        const fr a = r1cs_row(A, q, w, set, ~0ull, &unset), b = r1cs_row(B, q, w, set, ~0ull, &unset), rest = r1cs_row(C, q, w, set, g, &unset);
        if (unset) { atomicMin(&flags[1], (unsigned long long)(row_base + q)); continue; }
        r1cs_store(w, g, fr_sub(fr_mul(fr_to_mont(a), b), rest));
        set[g] = 1;
    }
}
__global__ void __launch_bounds__(256) r1cs_extras_kernel(DevMat A, DevMat B, DevMat C, size_t n_extra, u64 row_base, const u64 *__restrict__ w,
                                                          const unsigned char *__restrict__ set, u64 *__restrict__ a_ev, u64 *__restrict__ b_ev,
                                                          u64 *__restrict__ c_ev, unsigned long long *flags) {
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_extra) return;
    bool unset = false;
    const fr a = r1cs_row(A, q, w, set, ~0ull, &unset), b = r1cs_row(B, q, w, set, ~0ull, &unset), c = r1cs_row(C, q, w, set, ~0ull, &unset);
    if (unset) { atomicMin(&flags[1], (unsigned long long)(row_base + q)); return; }
    const fr ab = fr_mul(fr_to_mont(a), b);
    u32 diff = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) diff |= ab.l[i] ^ c.l[i];
    if (diff) atomicMin(&flags[0], (unsigned long long)(row_base + q));
    r1cs_store(a_ev, row_base + q, a); r1cs_store(b_ev, row_base + q, b); r1cs_store(c_ev, row_base + q, c);
}
// the rows of an arithmetic template: one lane per (instance, row); local wires resolve through the instance's input table
struct ArithDev { u64 n_in, n_int, n_rows, n_inst, first_row; const u64 *coef, *lc_ptr, *lc_ent, *rows, *inst; };
__device__ fr arith_lc_dev(const ArithDev &t, const u64 *in, u32 id, const u64 *w, const unsigned char *set, bool *unset) {
    auto global = [&](u64 local) { return local == 0 ? (u64)0 : local <= t.n_in ? in[local - 1] : in[t.n_in] + (local - 1 - t.n_in); };
    if (id & 0x80000000u) {
        const u64 g = global(id & 0x7FFFFFFFu);
        if (!set[g]) *unset = true;
        return fr_from_u64(w + 4 * g);
    }
    fr acc = fr_zero();
    for (u64 e = t.lc_ptr[id]; e < t.lc_ptr[id + 1]; e++) {
        const u64 g = global(t.lc_ent[e] & 0xFFFFFFFFull);
        if (!set[g]) *unset = true;
        acc = fr_add(acc, fr_mul(fr_to_mont(fr_from_u64(t.coef + 4 * (t.lc_ent[e] >> 32))), fr_from_u64(w + 4 * g)));
    }
    return acc;
}
__global__ void __launch_bounds__(256) r1cs_arith_rows_kernel(ArithDev t, const u64 *__restrict__ w, const unsigned char *__restrict__ set, u64 *__restrict__ a_ev,
                                                              u64 *__restrict__ b_ev, u64 *__restrict__ c_ev, unsigned long long *flags) {
    const size_t id = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (id >= t.n_inst * t.n_rows) return;
    // consecutive lanes take the SAME row of consecutive instances: they walk the same combination (no divergence) over different wires
    const u64 i = id % t.n_inst, q = id / t.n_inst;
    const u64 *in = t.inst + i * (t.n_in + 1);
    bool unset = false;
    const fr a = arith_lc_dev(t, in, (u32)(t.rows[2 * q] & 0xFFFFFFFFull), w, set, &unset), b = arith_lc_dev(t, in, (u32)(t.rows[2 * q] >> 32), w, set, &unset),
             c = arith_lc_dev(t, in, (u32)t.rows[2 * q + 1], w, set, &unset);
    const u64 row = t.first_row + i * t.n_rows + q;
    if (unset) { atomicMin(&flags[1], (unsigned long long)row); return; }
    const fr ab = fr_mul(fr_to_mont(a), b);
    u32 diff = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) diff |= ab.l[k] ^ c.l[k];
    if (diff) atomicMin(&flags[0], (unsigned long long)row);
    r1cs_store(a_ev, row, a); r1cs_store(b_ev, row, b); r1cs_store(c_ev, row, c);
}
__global__ void __launch_bounds__(256) r1cs_allset_kernel(const unsigned char *__restrict__ set, size_t n, unsigned long long *flags) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n && !set[j]) atomicMin(&flags[2], (unsigned long long)j);
}

#endif
}  // namespace

#ifndef ZP_R1CS_HOST_ONLY
// the circuit a ctx evaluated last, kept in HBM (the blob as it came, the list of its defining extra constraints); `checked`: the gadget template
// of the blob has been compared with the kernel on one instance (zp_r1cs_eval_device)
struct ZpG16Cache {
    std::vector<uint64_t> words;
    u64 *d_blob = nullptr, *d_defs = nullptr;
    size_t n_defs = 0;
    bool checked = false;
    // circuits with arithmetic templates: the host's copy of the witness (standard form; page-locked when the runtime grants it) the witness
    // programs run on, and its set flags -- kept between proofs, only the wires a proof sets are touched
    uint64_t *hW = nullptr;
    bool hW_pinned = false;
    std::vector<uint8_t> hset;
    std::vector<uint64_t> touched;       // the wires the last proof set from outside (their flags are cleared before the next one)
};
void zpi_g16_cache_free(zp_ctx *ctx) {
    if (!ctx->g16_cache) return;
    if (ctx->g16_cache->hW) { if (ctx->g16_cache->hW_pinned) (void)hipHostFree(ctx->g16_cache->hW); else free(ctx->g16_cache->hW); }
    if (ctx->g16_cache->d_blob) (void)hipFree(ctx->g16_cache->d_blob);
    if (ctx->g16_cache->d_defs) (void)hipFree(ctx->g16_cache->d_defs);
    delete ctx->g16_cache;
    ctx->g16_cache = nullptr;
}

namespace {

// device evaluation of the parsed circuit c (its blob resident at d_blob): d_w u64[>= n_wires][4] and d_set are overwritten; d_a / d_b / d_c
// u64[2^logm][4].  flags (host, 3 words): first violated row, first row that reads an unset wire, first wire left unset -- ~0 = none.
int32_t eval_device(zp_ctx *ctx, const Circ &c, const uint64_t *circ, const u64 *d_blob, const u64 *d_defs, size_t n_defs, const u64 *d_idx, const u64 *d_val,
                    size_t n_set, u64 *d_w, unsigned char *d_set, u64 *d_a, u64 *d_b, u64 *d_c, unsigned long long *d_flags, unsigned long long *h_flags,
                    const uint64_t *hW = nullptr, const std::function<int32_t()> &host_arith = nullptr) {
    const size_t m = (size_t)1 << c.logm;
    ZP_HIP(ctx, hipMemsetAsync(d_w, 0, c.n_wires * 32, ctx->stream));
    ZP_HIP(ctx, hipMemsetAsync(d_set, 0, c.n_wires, ctx->stream));
    ZP_HIP(ctx, hipMemsetAsync(d_a, 0, m * 32, ctx->stream));
    ZP_HIP(ctx, hipMemsetAsync(d_b, 0, m * 32, ctx->stream));
    ZP_HIP(ctx, hipMemsetAsync(d_c, 0, m * 32, ctx->stream));
    ZP_HIP(ctx, hipMemsetAsync(d_flags, 0xFF, 24, ctx->stream));
    hipLaunchKernelGGL(r1cs_scatter_kernel, dim3((unsigned)((n_set + 255) / 256)), dim3(256), 0, ctx->stream, d_idx, d_val, n_set, d_w, d_set);
    ZP_HIP(ctx, hipGetLastError());
    const u64 *d_inst = d_blob + (c.inst - circ);
    for (uint64_t wv = 0; wv < c.n_waves; wv++)
        ZP_TRY(zpi_r1cs_poseidon17(ctx, d_inst, c.waves[wv], c.waves[wv + 1] - c.waves[wv], d_w, d_set, d_a, d_b, d_c, d_flags, nullptr));
    DevMat E[3];
    for (int k = 0; k < 3; k++) E[k] = DevMat{d_blob + (c.E[k].ptr - circ), d_blob + (c.E[k].idx - circ), d_blob + (c.E[k].val - circ)};
    if (n_defs) {
        hipLaunchKernelGGL(r1cs_define_kernel, dim3(1), dim3(64), 0, ctx->stream, E[0], E[1], E[2], d_defs, n_defs, d_blob + (c.edef - circ), (u64)c.extra_base(), d_w,
                           d_set, d_flags);
        ZP_HIP(ctx, hipGetLastError());
    }
    if (c.n_extra) {
        hipLaunchKernelGGL(r1cs_extras_kernel, dim3((unsigned)((c.n_extra + 255) / 256)), dim3(256), 0, ctx->stream, E[0], E[1], E[2], (size_t)c.n_extra,
                           (u64)c.extra_base(), (const u64 *)d_w, (const unsigned char *)d_set, d_a, d_b, d_c, d_flags);
        ZP_HIP(ctx, hipGetLastError());
    }
    // The internal wires of the arithmetic templates come from the host (hW).  Their witness programs read caller-set wires only, and nothing
    // launched above reads an arithmetic wire: the programs run HERE, on the host, while the gadget instances and the explicit rows are on the
    // GPU (4.8 of the 13.6 ms witness step at the service's size were this host work in front of an idle GPU); every instance's wires then go up
    // as one contiguous range.
    if (!c.ar.empty()) {
        ZP_ARG(ctx, hW != nullptr, "internal: arithmetic templates without their host witness");
        if (host_arith) {
            const int32_t arc = host_arith();
            if (arc != ZP_OK) { (void)hipStreamSynchronize(ctx->stream); return arc; }
        }
        for (const ArithT &t : c.ar)
            for (uint64_t i = 0; i < t.n_inst; i++) {
                const uint64_t base = t.inst[i * (t.n_in + 1) + t.n_in];
                ZP_HIP(ctx, hipMemcpyAsync(d_w + 4 * base, hW + 4 * base, t.n_int * 32, hipMemcpyHostToDevice, ctx->stream));
                ZP_HIP(ctx, hipMemsetAsync(d_set + base, 1, t.n_int, ctx->stream));
            }
    }
    for (const ArithT &t : c.ar) {
        const ArithDev td = {t.n_in, t.n_int, t.n_rows, t.n_inst, t.first_row, d_blob + (t.coef - circ), d_blob + (t.lc_ptr - circ), d_blob + (t.lc_ent - circ),
                             d_blob + (t.rows - circ), d_blob + (t.inst - circ)};
        const size_t lanes = (size_t)(t.n_inst * t.n_rows);
        hipLaunchKernelGGL(r1cs_arith_rows_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, td, (const u64 *)d_w, (const unsigned char *)d_set, d_a,
                           d_b, d_c, d_flags);
        ZP_HIP(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(r1cs_allset_kernel, dim3((unsigned)((c.n_wires + 255) / 256)), dim3(256), 0, ctx->stream, (const unsigned char *)d_set, (size_t)c.n_wires, d_flags);
    ZP_HIP(ctx, hipGetLastError());
    return zpi_d2h_small(ctx, h_flags, d_flags, 24);
}

// the circuit in HBM (cached per ctx by content) + the one-time comparison of its gadget template with the kernel: ONE instance with arbitrary
// inputs through the host's generic evaluator (zp_r1cs_eval over the blob's own template matrices) and through the kernel
int32_t circuit_on_device(zp_ctx *ctx, const Circ &c, const uint64_t *circ, size_t words, ZpG16Cache **out) {
    ZpG16Cache *g = ctx->g16_cache;
    if (g && g->words.size() == words && memcmp(g->words.data(), circ, words * 8) == 0 && g->checked) { *out = g; return ZP_OK; }
    zpi_g16_cache_free(ctx);
    const uint64_t n_int = c.n_local - 1 - c.t;
    ZP_ARG(ctx, c.t == 17 && c.tc == n_int && c.tc >= 3 * 8 * 17 + 1 && (c.tc - 1 - 3 * 8 * 17) % 3 == 0,
           "the circuit's gadget is not the width-17 Poseidon permutation the device evaluator knows");
    g = new ZpG16Cache();
    ctx->g16_cache = g;
    g->words.assign(circ, circ + words);
    std::vector<u64> defs;
    for (uint64_t q = 0; q < c.n_extra; q++) if (c.edef[q] != ~0ull) defs.push_back(q);
    g->n_defs = defs.size();
    ZP_HIP(ctx, hipMalloc((void **)&g->d_blob, words * 8));
    ZP_HIP(ctx, hipMemcpyAsync(g->d_blob, circ, words * 8, hipMemcpyHostToDevice, ctx->stream));
    if (g->n_defs) {
        ZP_HIP(ctx, hipMalloc((void **)&g->d_defs, g->n_defs * 8));
        ZP_HIP(ctx, hipMemcpyAsync(g->d_defs, defs.data(), g->n_defs * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    ZP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // the mini circuit: the blob's template, one instance on wires 2..18 (wire 1: the public input, unused), no extras
    const size_t tpl_words = (size_t)(c.inst - circ) - 16;
    std::vector<uint64_t> mini(16, 0);
    uint64_t lm = 1;
    while ((1ull << lm) < c.tc + 1) lm++;
    const uint64_t nw = 2 + 17 + n_int;
    mini[0] = MAGIC; mini[1] = nw; mini[2] = c.tc; mini[3] = lm; mini[4] = 17; mini[5] = c.n_local; mini[6] = c.tc; mini[7] = 1; mini[8] = 0; mini[9] = 1; mini[10] = 1;
    mini.insert(mini.end(), circ + 16, circ + 16 + tpl_words);
    const size_t inst_at = mini.size();
    for (uint64_t k = 0; k < 17; k++) mini.push_back(2 + k);
    mini.push_back(19); mini.push_back(0);
    mini.push_back(0); mini.push_back(1);                  // waves
    for (int k = 0; k < 3; k++) mini.push_back(0);          // three empty matrices: ptr[1] = {0}
    Circ mc;
    if (!parse(mini.data(), mini.size(), &mc)) { ctx->err = "internal: mini circuit"; return ZP_ERR_INTERNAL; }
    const size_t mm = (size_t)1 << lm;
    std::vector<uint64_t> w(nw * 4, 0), ha(mm * 4), hb(mm * 4), hc(mm * 4), sidx(19), sval(19 * 4, 0);
    std::vector<uint8_t> set(nw, 0);
    w[0] = 1; set[0] = set[1] = 1;
    sidx[0] = 0; sval[0] = 1; sidx[1] = 1;
    for (uint64_t k = 0; k < 17; k++) {
        uint64_t *v = &w[(2 + k) * 4];
        for (int i = 0; i < 4; i++) v[i] = 0x9E3779B97F4A7C15ull * (k * 4 + i + 1) ^ (0xD1B54A32D192ED03ull >> (k + i));
        v[3] &= 0x0FFFFFFFFFFFFFFFull;                      // < r
        set[2 + k] = 1;
        sidx[2 + k] = 2 + k;
        memcpy(&sval[(2 + k) * 4], v, 32);
    }
    int64_t bad = -1;
    if (zp_r1cs_eval(mini.data(), mini.size(), w.data(), set.data(), ha.data(), hb.data(), hc.data(), &bad) != ZP_OK) {
        ctx->err = "the circuit's gadget template is not evaluable";
        return ZP_ERR_ARG;
    }
    void *d = nullptr;
    const size_t bytes = mini.size() * 8 + 19 * 40 + nw * 33 + 3 * mm * 32 + 64;
    ZP_TRY(zpi_pool_alloc(ctx, bytes + 64, &d));
    u64 *d_mini = (u64 *)d, *d_idx = d_mini + mini.size(), *d_val = d_idx + 19, *d_w = d_val + 19 * 4, *d_a = d_w + nw * 4, *d_b = d_a + mm * 4, *d_c = d_b + mm * 4;
    unsigned long long *d_flags = (unsigned long long *)(d_c + mm * 4);
    unsigned char *d_set = (unsigned char *)(d_flags + 4);
    unsigned long long hf[3];
    int32_t rc = ZP_OK;
    std::vector<uint64_t> ga(mm * 4), gb(mm * 4), gc(mm * 4), gw(nw * 4);
    if (hipMemcpyAsync(d_mini, mini.data(), mini.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_idx, sidx.data(), 19 * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_val, sval.data(), 19 * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        rc = ZP_ERR_HIP;
    if (rc == ZP_OK) rc = eval_device(ctx, mc, mini.data(), d_mini, nullptr, 0, d_idx, d_val, 19, d_w, d_set, d_a, d_b, d_c, d_flags, hf);
    if (rc == ZP_OK && (hipMemcpyAsync(ga.data(), d_a, mm * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipMemcpyAsync(gb.data(), d_b, mm * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipMemcpyAsync(gc.data(), d_c, mm * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipMemcpyAsync(gw.data(), d_w, nw * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = ZP_ERR_HIP;
    zpi_pool_release(ctx, d, bytes + 64);
    (void)inst_at;
    if (rc != ZP_OK) return rc;
    if (hf[0] != ~0ull || hf[1] != ~0ull || hf[2] != ~0ull || ga != ha || gb != hb || gc != hc || gw != w) {
        ctx->err = "the circuit's gadget template is not the width-17 Poseidon permutation of the installed tables (device evaluator refuses it)";
        return ZP_ERR_ARG;
    }
    g->checked = true;
    *out = g;
    return ZP_OK;
}

}  // namespace

extern "C" {

// zp_r1cs_eval on the GPU: the n_set caller-set wires in (host), the complete witness d_w u64[n_wires][4] and A w, B w, C w (d_a, d_b, d_c
// u64[2^logm][4]) out in HBM, the public inputs to the host.  Same results, same refusals (-20 / -21, *bad) as zp_r1cs_eval.  The gadget of the
// circuit must be the width-17 Poseidon permutation of the installed tables (compared with the kernel once per circuit and ctx).
int32_t zp_r1cs_eval_device(zp_ctx *ctx, const uint64_t *circ, size_t words, const uint64_t *set_idx, const uint64_t *set_val, size_t n_set, uint64_t *d_w,
                            uint64_t *d_a, uint64_t *d_b, uint64_t *d_c, uint64_t *out_pub, int64_t *bad) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    Circ c;
    ZP_ARG(ctx, parse(circ, words, &c), "malformed circuit blob");
    ZP_ARG(ctx, set_idx && set_val && d_w && d_a && d_b && d_c && out_pub && n_set >= 1 && n_set <= c.n_wires, "bad argument");
    if (bad) *bad = -1;
    bool one = false;
    for (size_t k = 0; k < n_set; k++) {
        ZP_ARG(ctx, set_idx[k] < c.n_wires && std_canonical(set_val + 4 * k), "a set wire is outside the circuit or its value is not below the modulus");
        if (set_idx[k] == 0) one = set_val[4 * k] == 1 && !(set_val[4 * k + 1] | set_val[4 * k + 2] | set_val[4 * k + 3]);
    }
    ZP_ARG(ctx, one, "wire 0 must be set to 1");
    ZpG16Cache *g = nullptr;
    ZP_TRY(circuit_on_device(ctx, c, circ, words, &g));
    void *d = nullptr;
    const size_t bytes = n_set * 40 + c.n_wires + 64;
    ZP_TRY(zpi_pool_alloc(ctx, bytes, &d));
    u64 *d_idx = (u64 *)d, *d_val = d_idx + n_set;
    unsigned long long *d_flags = (unsigned long long *)(d_val + 4 * n_set);
    unsigned char *d_set = (unsigned char *)(d_flags + 4);
    unsigned long long hf[3] = {~0ull, ~0ull, ~0ull};
    int32_t rc = ZP_OK;
    if (hipMemcpyAsync(d_idx, set_idx, n_set * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(d_val, set_val, n_set * 32, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
        ctx->err = "upload of the set wires failed";
        rc = ZP_ERR_HIP;
    }
    if (rc == ZP_OK && !c.ar.empty()) {
        // the witness programs of the arithmetic templates, on the host, over the caller-set wires (they read nothing else)
        if (!g->hW) {
            void *hp = nullptr;
            if (hipHostMalloc(&hp, c.n_wires * 32, hipHostMallocDefault) == hipSuccess) { g->hW = (uint64_t *)hp; g->hW_pinned = true; }
            else { (void)hipGetLastError(); g->hW = (uint64_t *)malloc(c.n_wires * 32); }
            if (!g->hW) { ctx->err = "out of host memory for the witness"; rc = ZP_ERR_NOMEM; }
            else g->hset.assign(c.n_wires, 0);
        }
        if (rc == ZP_OK) {
            for (uint64_t j : g->touched) g->hset[j] = 0;
            for (const ArithT &t : c.ar)
                for (uint64_t i = 0; i < t.n_inst; i++) memset(g->hset.data() + t.inst[i * (t.n_in + 1) + t.n_in], 0, t.n_int);
            g->touched.assign(set_idx, set_idx + n_set);
            for (size_t k = 0; k < n_set; k++) {
                memcpy(g->hW + 4 * set_idx[k], set_val + 4 * k, 32);
                g->hset[set_idx[k]] = 1;
            }
        }
    }
    // (the witness programs themselves run inside eval_device, while the GPU evaluates the gadget instances)
    auto host_arith = [&]() -> int32_t {
        const auto tw0 = std::chrono::steady_clock::now();
        const int32_t arc = arith_witness_all(c, g->hW, g->hset.data(), bad);
        if (getenv("ZP_R1CS_TIMING"))
            fprintf(stderr, "zp_r1cs_eval_device: witness programs of %zu arithmetic templates on the host: %.2f ms\n", c.ar.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
        if (arc != ZP_OK)
            ctx->err = arc == -20 ? "the assignment does not satisfy the circuit: no proof for a false statement" : "a wire of the circuit has no value";
        return arc;
    };
    if (rc == ZP_OK) rc = eval_device(ctx, c, circ, g->d_blob, g->d_defs, g->n_defs, d_idx, d_val, n_set, (u64 *)d_w, d_set, (u64 *)d_a, (u64 *)d_b, (u64 *)d_c, d_flags, hf,
                                      g->hW, c.ar.empty() ? std::function<int32_t()>() : std::function<int32_t()>(host_arith));
    if (rc == ZP_OK) rc = zpi_d2h_small(ctx, out_pub, (u64 *)d_w + 4, c.n_pub * 32);
    zpi_pool_release(ctx, d, bytes);
    if (rc != ZP_OK) return rc;
    if (hf[1] != ~0ull || hf[2] != ~0ull) {
        if (bad) *bad = (int64_t)(hf[1] != ~0ull ? hf[1] : hf[2]);
        ctx->err = "a wire of the circuit has no value";
        return -21;
    }
    if (hf[0] != ~0ull) {
        if (bad) *bad = (int64_t)hf[0];
        ctx->err = "the assignment does not satisfy the circuit: no proof for a false statement";
        return -20;
    }
    return ZP_OK;
}

}  // extern "C"

#endif
namespace {

}  // namespace

extern "C" {

// script: [0] "PZWRAPS1" [1] entries [2] wires set in all [3] n_queries [4] n_trees [5] index bits, per tree (width, leaves, levels), then per entry six
// words: op, first wire, count, a, b, c --
//   0 constant a                      1 aux                                  2 root of tree b               3 index of query a
//   4 bits 0..count-1 of that index   5 elements of block c of the leaf (query a, tree b)                    6 the 16 digests of level c on the path
//   7 one-hot of the position at level c (16)       8 / 9 one-hot of its low / high two bits (4)
//   10 elements b .. b + count - 1 of absorbed block a of the transcript            11 the 16 rate elements of squeeze permutation a
//   12 bits c .. c + count - 1 of rate element b of squeeze permutation a           13 entries c .. of that element's "equal to r so far" chain
//      (walking down from bit 253 over the positions where r has a 1: the AND of the element's bits there, first position excluded)
//   14 the 30 partial products of the top 32 bits of 64-bit word c of that element (bits 32..33, 32..34, ... 32..62 of the word)
//   15 the capacity element after permutation a of the transcript (absorbing permutations in order, then the squeeze-only ones)
//   16 challenge element a (rate element 1 after absorbed segment a)               17 its low 192 bits (zeta as the circuit commits to it)
// aux u64[n_aux][4]: element 0 the value the proof is bound to (the aggregator address), elements 1.. whatever else the circuit takes from its
// caller (stage B-2: the statement's sparse fixed columns at zeta, one packed element each) -- op 1 reads element a.
// out_idx u64[cap], out_val u64[cap][4] (standard form) receive the wires and their values; *n_set their number ([2] of the script).
int32_t zp_wrap_assign(const uint64_t *script, size_t script_words, const uint64_t *openings, size_t open_words, const uint64_t *aux, size_t n_aux, uint64_t *out_idx,
                       uint64_t *out_val, size_t cap, size_t *n_set) {
    Openings o;
    if (!script || script_words < 6 || script[0] != SCRIPT_MAGIC || !aux || n_aux < 1 || !out_idx || !out_val || !n_set || !parse_openings(openings, open_words, &o))
        return ZP_ERR_ARG;
    const uint64_t ne = script[1], total = script[2];
    if (script[3] != o.nq || script[4] != o.ntr || script[5] != o.logm || ne > (1ull << 28) || script_words != 8 + 3 * o.ntr + 6 * ne || total > cap)
        return ZP_ERR_ARG;
    for (size_t i = 0; i < n_aux; i++) if (!std_canonical(aux + 4 * i)) return ZP_ERR_ARG;
    for (uint64_t i = 0; i < o.n_chal; i++) if (!std_canonical(o.chal + 4 * i)) return ZP_ERR_ARG;
    for (uint64_t t = 0; t < o.ntr; t++)
        if (script[6 + 3 * t] != o.tr[t].width || script[7 + 3 * t] != o.tr[t].leaves || script[8 + 3 * t] != o.tr[t].levels) return ZP_ERR_ARG;   // another layout
    if (script[6 + 3 * o.ntr] != o.n_blocks || script[7 + 3 * o.ntr] != o.n_rates) return ZP_ERR_ARG;                                              // another transcript
    for (uint64_t i = 0; i < (o.n_blocks + o.n_rates) * 16 + (o.n_blocks + o.n_rates - 1); i++)
        if (!std_canonical(o.blocks + 4 * i)) return ZP_ERR_ARG;
    // positions of r's one-bits below the top one, walking down: the chain of op 13
    unsigned ones[254], n_ones = 0;
    for (int i = 252; i >= 0; i--)
        if (bit_of(FR_R, (unsigned)i)) ones[n_ones++] = (unsigned)i;
    const uint64_t *e = script + 8 + 3 * o.ntr;
    size_t n = 0;
    for (uint64_t k = 0; k < ne; k++, e += 6) {
        const uint64_t op = e[0], wire = e[1], cnt = e[2], a = e[3], b = e[4], c = e[5];
        if (cnt < 1 || cnt > 64 || n + cnt > total || op > 17) return ZP_ERR_ARG;
        if (op >= 16) {                      // a challenge element, whole (16) or its low 192 bits (17)
            if (cnt != 1 || a >= o.n_chal) return ZP_ERR_ARG;
            out_idx[n] = wire;
            memcpy(out_val + 4 * n, o.chal + 4 * a, 32);
            if (op == 17) out_val[4 * n + 3] = 0;
            n++;
            continue;
        }
        if (op == 15) {                      // the capacity after permutation a of the transcript
            if (cnt != 1 || a >= o.n_blocks + o.n_rates - 1) return ZP_ERR_ARG;
            out_idx[n] = wire;
            memcpy(out_val + 4 * n, o.caps + 4 * a, 32);
            n++;
            continue;
        }
        if (op >= 10) {
            // (sums of script words are compared WITHOUT forming them: b + cnt wraps for a b near 2^64 -- found by tests/test_r1cs_fuzz.py, round 6)
            if (op == 10 ? (a >= o.n_blocks || b >= 16 || cnt > 16 - b) : a >= o.n_rates) return ZP_ERR_ARG;
            if ((op == 11 && cnt != 16) || (op >= 12 && b >= 16) || (op == 12 && (c >= 254 || cnt > 254 - c)) || (op == 13 && (c >= n_ones || cnt > n_ones - c)) ||
                (op == 14 && (cnt != 30 || c >= 3)))
                return ZP_ERR_ARG;
            const uint64_t *el = op == 10 ? o.blocks + (a * 16 + b) * 4 : o.rates + (a * 16 + (op == 11 ? 0 : b)) * 4;
            for (uint64_t i = 0; i < cnt; i++, n++) {
                uint64_t *v = out_val + 4 * n;
                out_idx[n] = wire + i;
                v[0] = v[1] = v[2] = v[3] = 0;
                if (op <= 11) memcpy(v, el + 4 * i, 32);
                else if (op == 12) v[0] = bit_of(el, (unsigned)(c + i));
                else if (op == 13) {
                    uint64_t p = bit_of(el, 253);
                    for (uint64_t t = 0; t <= c + i; t++) p &= bit_of(el, ones[t]);
                    v[0] = p;
                } else {
                    uint64_t p = 1;
                    for (uint64_t t = 0; t <= i + 1; t++) p &= bit_of(el, (unsigned)(64 * c + 32 + t));
                    v[0] = p;
                }
            }
            continue;
        }
        if (op >= 3 && a >= o.nq) return ZP_ERR_ARG;
        if (op == 1 && a >= n_aux) return ZP_ERR_ARG;
        if ((op == 2 || op >= 5) && b >= o.ntr) return ZP_ERR_ARG;
        const uint64_t *q = op >= 3 ? o.query(a) : nullptr;
        const OpenTree *T = (op == 2 || op >= 5) ? &o.tr[b] : nullptr;
        uint64_t pos = 0;
        if (op >= 6) {
            if (c >= T->levels) return ZP_ERR_ARG;
            pos = ((q[0] & (T->leaves - 1)) >> (4 * c)) & 15;
        }
        if ((op == 5 && (cnt != 16 || c >= (T->width + 55) / 56)) || ((op == 6 || op == 7) && cnt != 16) || (op >= 8 && cnt != 4) || (op == 4 && cnt > 64) ||
            ((op == 1 || op == 2 || op == 3) && cnt != 1))
            return ZP_ERR_ARG;
        for (uint64_t i = 0; i < cnt; i++, n++) {
            uint64_t *v = out_val + 4 * n;
            out_idx[n] = wire + i;
            v[0] = v[1] = v[2] = v[3] = 0;
            switch (op) {
                case 0: v[0] = a; break;
                case 1: memcpy(v, aux + 4 * a, 32); break;
                case 2: memcpy(v, T->root, 32); break;
                case 3: v[0] = q[0]; break;
                case 4: v[0] = (q[0] >> i) & 1; break;
                case 5: pack_element(q + T->off, T->width, c, (int)i, v); break;
                case 6: memcpy(v, q + T->off + T->width + (c * 16 + i) * 4, 32); break;
                case 7: v[0] = pos == i; break;
                case 8: v[0] = (pos & 3) == i; break;
                default: v[0] = (pos >> 2) == i; break;
            }
        }
    }
    if (n != total) return ZP_ERR_ARG;
    *n_set = n;
    return ZP_OK;
}

// The aux list of a stage B-2 wrap (zp_wrap_assign's `aux`) from what the host of GenFinalProof holds: the final STARK's openings record (its
// challenge element 1 IS zeta: three low 64-bit words, each mod p), the final STARK's statement (constraint program, public inputs, trace length,
// root of unity) and the element the proof is bound to.  out_aux u64[cap][4]: element 0 = addr4, element 1 + k = sparse fixed column k of the
// program at zeta, components packed c0 + c1 2^64 + c2 2^128; *n_aux = n_fixed - 1.  zeta3_out (may be NULL) u64[3]: the three WORDS zeta is read from.
int32_t zp_wrap_aux(const uint64_t *openings, size_t open_words, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub, int32_t logn,
                    uint64_t root32, const uint64_t *addr4, uint64_t *out_aux, size_t cap, size_t *n_aux, uint64_t *zeta3_out) {
    Openings o;
    if (!h_program || program_words < 12 || !addr4 || !out_aux || !n_aux || !parse_openings(openings, open_words, &o) || o.n_chal < 2 || !std_canonical(addr4))
        return ZP_ERR_ARG;
    const uint64_t n_fixed = h_program[3];
    if (n_fixed < 2 || n_fixed > 4096 || cap < n_fixed - 1) return ZP_ERR_ARG;
    uint64_t zeta[3];
    for (int k = 0; k < 3; k++) zeta[k] = o.chal[4 + k] % GLP;
    try {
        std::vector<uint64_t> fixed(3 * n_fixed);
        const int32_t rc = zp_program_fixed_eval_ext(h_program, program_words, h_pub, n_pub, logn, root32, zeta, fixed.data(), (int32_t)n_fixed, 0);
        if (rc != ZP_OK) return rc;
        memcpy(out_aux, addr4, 32);
        for (uint64_t k = 2; k < n_fixed; k++) {
            uint64_t *e = out_aux + 4 * (k - 1);
            e[0] = fixed[3 * k]; e[1] = fixed[3 * k + 1]; e[2] = fixed[3 * k + 2]; e[3] = 0;
        }
        *n_aux = n_fixed - 1;
        if (zeta3_out) memcpy(zeta3_out, o.chal + 4, 24);       // the words as the circuit commits to them (a word >= p names its residue a second time)
        return ZP_OK;
    } catch (...) {
        return ZP_ERR_NOMEM;
    }
}

#ifndef ZP_R1CS_HOST_ONLY
// One Groth16 proof.  circ: the circuit blob; d_u1x (G1, u32[n_wires + 2][16]): [u_j]_1 | alpha_1 | delta_1; d_v_wires u32[n_v]: the wires with a
// non-zero column in B, ascending (a third of the gadget's wires never stand in B: their key points would be infinity); d_v1x (G1, u32[n_v + 2][16]):
// [v_j]_1 of those wires | beta_1 | delta_1; d_v2x (G2, u32[n_v + 2][32]): [v_j]_2 of those wires | beta_2 | delta_2; d_l1 (G1, u32[n_wires][16], infinity at wire 0 and the public inputs: those are
// the verifier's); d_h1 (G1, u32[2^logm - 1][16]) -- device-resident key points in the MSM layout; h_delta1 u32[16].
// set_idx / set_val: the n_set caller-set wires (zp_wrap_assign; wire 0 = 1 among them).  h_r, h_s: the blinding scalars (4 words each, standard
// form).  out_a u32[16], out_b u32[32], out_c u32[16]: pi_a, pi_b, pi_c (affine, standard form); out_pub u64[n_pub][4]: the public inputs the proof
// is for; h_ms (may be NULL) double[8]: milliseconds of witness completion, QAP step, all MSMs, then the MSMs one by one (A, B in G1, B in G2, l, h).  -20 / -21 as zp_r1cs_eval (*bad): no proof for a
// false statement.
int32_t zp_groth16_prove(zp_ctx *ctx, const uint64_t *circ, size_t words, const uint32_t *d_u1x, const uint32_t *d_v_wires, size_t n_v, const uint32_t *d_v1x,
                         const uint32_t *d_v2x, const uint32_t *d_l1, const uint32_t *d_h1, const uint32_t *h_delta1, const uint64_t *set_idx, const uint64_t *set_val, size_t n_set,
                         const uint64_t *h_r, const uint64_t *h_s, uint32_t *out_a, uint32_t *out_b, uint32_t *out_c, uint64_t *out_pub, double *h_ms,
                         int64_t *bad) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_BIND(ctx);
    Circ c;
    ZP_ARG(ctx, parse(circ, words, &c), "malformed circuit blob");
    ZP_ARG(ctx, d_u1x && d_v_wires && n_v >= 1 && n_v <= c.n_wires && d_v1x && d_v2x && d_l1 && d_h1 && h_delta1 && set_idx && set_val && h_r && h_s && out_a && out_b && out_c && out_pub, "null argument");
    ZP_ARG(ctx, std_canonical(h_r) && std_canonical(h_s), "blinding scalars must be below the group order");
    if (bad) *bad = -1;
    const size_t n = c.n_wires, m = (size_t)1 << c.logm;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    void *d_abc = nullptr, *d_sc = nullptr, *d_tail = nullptr;
    const size_t abc_bytes = 3 * m * 32, sc_bytes = (n + 2 + n_v + 2) * 32, tail_bytes = 5 * (64 + 32);
    int32_t rc = ZP_OK;
    try {
        ZP_TRY(zpi_pool_alloc(ctx, abc_bytes, &d_abc));
        rc = zpi_pool_alloc(ctx, sc_bytes, &d_sc);
        if (rc == ZP_OK) rc = zpi_pool_alloc(ctx, tail_bytes, &d_tail);
        auto done = [&](int32_t r) {
            if (d_abc) zpi_pool_release(ctx, d_abc, abc_bytes);
            if (d_sc) zpi_pool_release(ctx, d_sc, sc_bytes);
            if (d_tail) zpi_pool_release(ctx, d_tail, tail_bytes);
            return r;
        };
        if (rc != ZP_OK) return done(rc);
        uint64_t *da = (uint64_t *)d_abc, *db = da + 4 * m, *dc = db + 4 * m;
        // the witness is completed where the MSMs read it: [w | 1 | r or s] (two extra scalars for the blinding terms that ride in the MSMs)
        if ((rc = zp_r1cs_eval_device(ctx, circ, words, set_idx, set_val, n_set, (uint64_t *)d_sc, da, db, dc, out_pub, bad)) != ZP_OK) return done(rc);
        uint64_t ext[8] = {1, 0, 0, 0, h_r[0], h_r[1], h_r[2], h_r[3]};
        if ((rc = zpi_h2d_small(ctx, (uint64_t *)d_sc + 4 * n, ext, 64)) != ZP_OK) return done(rc);
        uint64_t *d_scv = (uint64_t *)d_sc + 4 * (n + 2);                          // [w_j of the wires in B | 1 | s]
        hipLaunchKernelGGL(r1cs_gather_kernel, dim3((unsigned)((n_v + 255) / 256)), dim3(256), 0, ctx->stream, (const u64 *)d_sc, d_v_wires, n_v, (u64 *)d_scv);
        memcpy(ext + 4, h_s, 32);
        if ((rc = zpi_h2d_small(ctx, d_scv + 4 * n_v, ext, 64)) != ZP_OK) return done(rc);
        const auto t1 = now();
        // H = (A B - C) / Z: its coefficients replace A's evaluations and are the scalars of the h MSM
        const uint64_t coset[4] = {7, 0, 0, 0};
        if ((rc = zp_qap_quotient_bn254(ctx, da, db, dc, (int32_t)c.logm, coset)) != ZP_OK) return done(rc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "QAP step failed"; return done(ZP_ERR_HIP); }
        const auto t2 = now();
        uint32_t A1[16], B1[16], Cl[16], Ch[16];
        const uint32_t *sc = (const uint32_t *)d_sc;
        // The five MSMs are independent and, at 1-2 M points, each is a chain of short launches with two host round trips: they run on five
        // streams at once -- A on this ctx, the others on helper ctxs of their own (stream + Pippenger arena, kept for the life of the ctx);
        // everything they read was finished by the synchronisation above.  Knob g16_parallel (0: one after the other).
        int32_t rcs[5] = {ZP_OK, ZP_OK, ZP_OK, ZP_OK, ZP_OK};
        double tms[5] = {0, 0, 0, 0, 0};
        zp_ctx *hc[5] = {ctx, ctx, ctx, ctx, ctx};
        bool par = ctx->tune_g16_parallel != 0;
        for (int i = 0; par && i < 4; i++) {
            if (!ctx->msm_helpers[i] && zp_create(&ctx->msm_helpers[i], ctx->device) != ZP_OK) par = false;
            if (par) {
                ctx->msm_helpers[i]->tune_msm_c = ctx->tune_msm_c;
                ctx->msm_helpers[i]->tune_msm_chunk_log = ctx->tune_msm_chunk_log;
                hc[i + 1] = ctx->msm_helpers[i];
            }
        }
        if (!par) for (int i = 1; i < 5; i++) hc[i] = ctx;
        auto job = [&](int k) noexcept {       // runs on helper threads: nothing may escape (an exception there would end the process)
          try {
            const auto ta = now();
            switch (k) {
                case 0: rcs[0] = zp_msm_bn254(hc[0], d_u1x, sc, n + 2, A1); break;                                  // alpha + sum_j w_j u_j + r delta
                case 1: rcs[1] = zp_msm_bn254(hc[1], d_v1x, (const uint32_t *)d_scv, n_v + 2, B1); break;          // beta + sum_j w_j v_j + s delta
                case 2: rcs[2] = zp_msm_bn254_g2(hc[2], d_v2x, (const uint32_t *)d_scv, n_v + 2, out_b); break;
                case 3: rcs[3] = zp_msm_bn254(hc[3], d_l1, sc, n, Cl); break;
                default: rcs[4] = zp_msm_bn254(hc[4], d_h1, (const uint32_t *)da, m - 1, Ch); break;               // sum_i H_i [tau^i Z(tau) / delta]
            }
            tms[k] = ms(ta, now());
          } catch (...) {
            rcs[k] = ZP_ERR_INTERNAL;
          }
        };
        if (par) {
            // a thread that cannot be started (EAGAIN) is not fatal: the ones already running are joined, the remaining MSMs run here
            std::thread th[4];
            int started = 0;
            try {
                for (int k = 1; k < 5; k++) { th[k - 1] = std::thread(job, k); started = k; }
            } catch (...) {
            }
            job(0);
            for (int k = started + 1; k < 5; k++) { hc[k] = ctx; job(k); }
            for (int k = 1; k <= started; k++) th[k - 1].join();
        } else {
            for (int k = 0; k < 5; k++) job(k);
        }
        for (int k = 0; k < 5; k++)
            if (rcs[k] != ZP_OK) {
                if (rcs[k] == ZP_ERR_INTERNAL && hc[k]->err.empty()) ctx->err = "exception in an MSM of zp_groth16_prove";
                else if (hc[k] != ctx) ctx->err = hc[k]->err;
                return done(rcs[k]);
            }
        const auto t7 = now();
        // pi_c = Cl + Ch + s A + r B1 - r s delta: one more (five-point) MSM
        const Fr fr_r = fr_from_std(h_r), fr_s = fr_from_std(h_s);
        const Fr zero = {{0, 0, 0, 0}};
        uint64_t tsc[5][4] = {{1, 0, 0, 0}, {1, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
        memcpy(tsc[2], h_s, 32);
        memcpy(tsc[3], h_r, 32);
        fr_to_std(fr_sub(zero, fr_mul(fr_r, fr_s)), tsc[4]);
        uint32_t tpt[5][16];
        memcpy(tpt[0], Cl, 64); memcpy(tpt[1], Ch, 64); memcpy(tpt[2], A1, 64); memcpy(tpt[3], B1, 64); memcpy(tpt[4], h_delta1, 64);
        if ((rc = zpi_h2d_small(ctx, d_tail, tpt, sizeof tpt)) != ZP_OK) return done(rc);
        if ((rc = zpi_h2d_small(ctx, (uint8_t *)d_tail + sizeof tpt, tsc, sizeof tsc)) != ZP_OK) return done(rc);
        if ((rc = zp_msm_bn254(ctx, (const uint32_t *)d_tail, (const uint32_t *)((uint8_t *)d_tail + sizeof tpt), 5, out_c)) != ZP_OK) return done(rc);
        memcpy(out_a, A1, 64);
        if (h_ms) {
            h_ms[0] = ms(t0, t1); h_ms[1] = ms(t1, t2); h_ms[2] = ms(t2, now());
            for (int k = 0; k < 5; k++) h_ms[3 + k] = tms[k];       // each MSM on its own clock (they overlap; h_ms[2] is their wall time + the tail)
            (void)t7;
        }
        return done(ZP_OK);
    } catch (...) {
        if (d_abc) zpi_pool_release(ctx, d_abc, abc_bytes);
        if (d_sc) zpi_pool_release(ctx, d_sc, sc_bytes);
        if (d_tail) zpi_pool_release(ctx, d_tail, tail_bytes);
        ctx->err = "out of host memory in zp_groth16_prove";
        return ZP_ERR_NOMEM;
    }
}

#endif
}  // extern "C"
