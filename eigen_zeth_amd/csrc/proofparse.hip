// The query openings of a recursive-proof TEXT as arrays, without the interpreter: what GenAggregatedProof / GenFinalProof do first with the
// proofs a client hands in (proto/prover/v1/prover.proto:115-148: recursive_proof_1 / recursive_proof_2 / recursive_proof are strings;
// src/prover/provider.rs:422-433,472-483 sends them verbatim).  A chunk proof is ~1.2 MB of decimal numbers, 97 % of it inside "queries";
// a general-purpose JSON parser building one object per number spent 8 ms per proof there.  Here: one pass that skips to a
// key of the top-level object (zp_json_key_span), one pass that sizes the openings (zp_proof_queries_scan), one that writes them
// (zp_proof_queries_parse).  The small remainder (the header: parameters, roots, evaluations, final layer) stays with the host's own JSON
// reader.  Host code; strict: anything but the grammar stark/prover.py and csrc/prove.hip write is an error, and the caller falls back to
// its general parser (which then reports what is wrong with the text).
#include <cstdint>
#include <cstring>

#include "ctx.hpp"

namespace {

struct Cur {
    const char *p, *end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) p++; }
    bool eat(char c) {
        ws();
        if (p < end && *p == c) { p++; return true; }
        return false;
    }
    bool need(char c) {
        if (!eat(c)) ok = false;
        return ok;
    }
    bool peek(char c) { ws(); return p < end && *p == c; }
    // a JSON string without escapes we care about: returns [b, e) of its content
    bool str(const char **b, const char **e) {
        ws();
        if (p >= end || *p != '"') { ok = false; return false; }
        p++;
        *b = p;
        while (p < end && *p != '"') {
            if (*p == '\\') { p++; if (p >= end) break; }
            p++;
        }
        if (p >= end) { ok = false; return false; }
        *e = p;
        p++;
        return true;
    }
    bool key_is(const char *b, const char *e, const char *k) { return (size_t)(e - b) == strlen(k) && memcmp(b, k, e - b) == 0; }
    bool u64v(uint64_t *out) {                 // non-negative decimal integer < 2^64
        ws();
        if (p >= end || *p < '0' || *p > '9') { ok = false; return false; }
        uint64_t v = 0;
        int digits = 0;
        while (p < end && *p >= '0' && *p <= '9') {
            const unsigned d = (unsigned)(*p - '0');
            if (v > (UINT64_MAX - d) / 10) { ok = false; return false; }
            v = v * 10 + d;
            p++;
            digits++;
        }
        if (p < end && (*p == '.' || *p == 'e' || *p == 'E')) { ok = false; return false; }
        *out = v;
        return digits > 0;
    }
    void skip_value(int depth = 0) {          // any JSON value
        ws();
        if (!ok || p >= end || depth > 64) { ok = false; return; }
        if (*p == '"') { const char *b, *e; str(&b, &e); return; }
        if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            const bool obj = *p == '{';
            p++;
            if (eat(close)) return;
            for (;;) {
                if (obj) { const char *b, *e; if (!str(&b, &e) || !need(':')) return; }
                skip_value(depth + 1);
                if (!ok) return;
                if (eat(',')) continue;
                need(close);
                return;
            }
        }
        const char *q = p;                    // number / true / false / null
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') p++;
        if (p == q) ok = false;
    }
};

// walks the members of the object at c; calls f(key begin, key end) positioned at the value; f must consume the value
template <typename F>
bool each_member(Cur &c, F f) {
    if (!c.need('{')) return false;
    if (c.eat('}')) return true;
    for (;;) {
        const char *b, *e;
        if (!c.str(&b, &e) || !c.need(':')) return false;
        f(b, e);
        if (!c.ok) return false;
        if (c.eat(',')) continue;
        return c.need('}');
    }
}

constexpr int MAX_TREES = 48;

struct Sizes { int n_queries = 0, has_stage2 = 0, n_fri = 0, w[MAX_TREES] = {}, d[MAX_TREES] = {}; };

// one opening {"values":[..],"path":[[4]..]}: counts (vals == nullptr) or writes
bool opening(Cur &c, int *w, int *depth, uint64_t *vals, uint64_t *path) {
    int nw = 0, nd = 0;
    bool got_v = false, got_p = false;
    each_member(c, [&](const char *b, const char *e) {
        if (c.key_is(b, e, "values")) {
            if (got_v) { c.ok = false; return; }   // a repeated key is not this grammar (and must not restart a count)
            got_v = true;
            if (!c.need('[')) return;
            if (!c.eat(']'))
                for (;;) {
                    uint64_t v;
                    if (!c.u64v(&v)) return;
                    if (vals) { if (nw >= *w) { c.ok = false; return; } vals[nw] = v; }
                    nw++;
                    if (c.eat(',')) continue;
                    c.need(']');
                    break;
                }
        } else if (c.key_is(b, e, "path")) {
            if (got_p) { c.ok = false; return; }
            got_p = true;
            if (!c.need('[')) return;
            if (!c.eat(']'))
                for (;;) {
                    if (!c.need('[')) return;
                    for (int k = 0; k < 4; k++) {
                        uint64_t v;
                        if (!c.u64v(&v)) return;
                        if (path) { if (nd >= *depth) { c.ok = false; return; } path[4 * nd + k] = v; }
                        if (k < 3 && !c.need(',')) return;
                    }
                    if (!c.need(']')) return;
                    nd++;
                    if (c.eat(',')) continue;
                    c.need(']');
                    break;
                }
        } else {
            c.skip_value();
        }
    });
    if (!c.ok || !got_v || !got_p) return c.ok = false;
    if (vals) return c.ok = (nw == *w && nd == *depth);
    *w = nw;
    *depth = nd;
    return true;
}

// one query object.  Tree order of the outputs: trace, [stage2], quotient, fri0, fri1, ...  (stark/verifier_air.py Shape.trees)
bool query(Cur &c, Sizes &sz, bool first, uint64_t *index, uint64_t **vals, uint64_t **paths) {
    bool got_i = false, got_t = false, got_q = false, got_s = false, got_f = false;
    Sizes me;
    each_member(c, [&](const char *b, const char *e) {
        auto one = [&](int t) {
            if (t < 0 || t >= MAX_TREES) { c.ok = false; return; }
            if (vals) {
                // write mode: only the trees the caller sized exist (their pointers are the only initialised ones)
                if (t >= 2 + sz.has_stage2 + sz.n_fri) { c.ok = false; return; }
                int w = sz.w[t], d = sz.d[t];
                opening(c, &w, &d, vals[t], paths[t]);
            }
            else opening(c, &me.w[t], &me.d[t], nullptr, nullptr);
        };
        const int tq = vals ? 1 + sz.has_stage2 : 2;        // while sizing, slot 1 is kept free for a stage-2 opening (compacted below)
        auto once = [&](bool &got) { if (got) c.ok = false; got = true; return c.ok; };   // a repeated key is an error in both passes
        if (c.key_is(b, e, "index")) { if (!once(got_i)) return; uint64_t v; if (c.u64v(&v) && index) *index = v; }
        else if (c.key_is(b, e, "trace")) { if (!once(got_t)) return; one(0); }
        else if (c.key_is(b, e, "stage2")) { if (!once(got_s)) return; if (vals && !sz.has_stage2) { c.ok = false; return; } one(1); }
        else if (c.key_is(b, e, "quotient")) { if (!once(got_q)) return; one(tq); }
        else if (c.key_is(b, e, "fri")) {
            if (!once(got_f)) return;
            int li = 0;
            if (!c.need('[')) return;
            if (!c.eat(']'))
                for (;;) {
                    if (tq + 1 + li >= MAX_TREES) { c.ok = false; return; }
                    one(tq + 1 + li);
                    if (!c.ok) return;
                    li++;
                    if (c.eat(',')) continue;
                    c.need(']');
                    break;
                }
            me.n_fri = li;
        } else {
            c.skip_value();
        }
    });
    if (!c.ok || !got_i || !got_t || !got_q || !got_f) return c.ok = false;
    if (vals) return c.ok = (got_s == (sz.has_stage2 != 0)) && me.n_fri == sz.n_fri;
    me.has_stage2 = got_s ? 1 : 0;
    // compact: without a stage-2 opening the trees behind slot 1 move down
    if (!got_s)
        for (int t = 1; t < MAX_TREES - 1; t++) { me.w[t] = me.w[t + 1]; me.d[t] = me.d[t + 1]; }
    const int T = 2 + me.has_stage2 + me.n_fri;
    if (first) {
        sz.has_stage2 = me.has_stage2; sz.n_fri = me.n_fri;
        memcpy(sz.w, me.w, sizeof me.w); memcpy(sz.d, me.d, sizeof me.d);
    } else {
        if (me.has_stage2 != sz.has_stage2 || me.n_fri != sz.n_fri) return c.ok = false;
        for (int t = 0; t < T; t++)
            if (me.w[t] != sz.w[t] || me.d[t] != sz.d[t]) return c.ok = false;      // ragged openings
    }
    return true;
}

}  // namespace

extern "C" {

// [begin, end) of the VALUE of member `key` of the JSON object that starts at text[0 ..): ZP_OK, or ZP_ERR_ARG when the text is not an
// object / has no such member.
int32_t zp_json_key_span(const char *text, size_t len, const char *key, size_t *begin, size_t *end) {
    if (!text || !key || !begin || !end) return ZP_ERR_ARG;
    Cur c{text, text + len};
    bool found = false;
    each_member(c, [&](const char *b, const char *e) {
        c.ws();
        const char *v0 = c.p;
        c.skip_value();
        if (c.ok && !found && c.key_is(b, e, key)) { found = true; *begin = (size_t)(v0 - text); *end = (size_t)(c.p - text); }
    });
    return c.ok && found ? ZP_OK : ZP_ERR_ARG;
}

// Sizes of the openings in the "queries" array of the proof object text[0 .. len): where the array sits ([q_begin, q_end): the caller cuts
// it out and reads the rest with its own JSON parser), the number of queries, whether there is a stage-2 tree, the number of FRI layers, and
// per tree (trace, [stage2], quotient, fri0 ..) the leaf width and the path length -- every query must have the same.  ZP_ERR_ARG: not
// this grammar.
int32_t zp_proof_queries_scan(const char *text, size_t len, size_t *q_begin, size_t *q_end, int32_t *n_queries, int32_t *has_stage2, int32_t *n_fri,
                              int32_t *widths, int32_t *depths, int32_t max_trees) {
    if (!text || !q_begin || !q_end || !n_queries || !has_stage2 || !n_fri || !widths || !depths) return ZP_ERR_ARG;
    size_t b = 0, e = 0;
    if (zp_json_key_span(text, len, "queries", &b, &e) != ZP_OK) return ZP_ERR_ARG;
    Cur c{text + b, text + e};
    Sizes sz;
    int nq = 0;
    if (!c.need('[')) return ZP_ERR_ARG;
    if (!c.eat(']'))
        for (;;) {
            if (!query(c, sz, nq == 0, nullptr, nullptr, nullptr)) return ZP_ERR_ARG;
            nq++;
            if (c.eat(',')) continue;
            if (!c.need(']')) return ZP_ERR_ARG;
            break;
        }
    const int T = 2 + sz.has_stage2 + sz.n_fri;
    if (nq < 1 || T > max_trees || T > MAX_TREES) return ZP_ERR_ARG;
    *q_begin = b; *q_end = e; *n_queries = nq; *has_stage2 = sz.has_stage2; *n_fri = sz.n_fri;
    for (int t = 0; t < T; t++) { widths[t] = sz.w[t]; depths[t] = sz.d[t]; }
    return ZP_OK;
}

// The openings themselves.  index u64[nq]; values: per tree t a block u64[nq][w_t], blocks one after the other; paths: per tree a block
// u64[nq][d_t][4].  The sizes are those zp_proof_queries_scan reported for this text.
int32_t zp_proof_queries_parse(const char *text, size_t q_begin, size_t q_end, int32_t n_queries, int32_t has_stage2, int32_t n_fri,
                               const int32_t *widths, const int32_t *depths, uint64_t *index, uint64_t *values, uint64_t *paths) {
    if (!text || !widths || !depths || !index || !values || !paths || n_queries < 1 || q_end <= q_begin || n_fri < 0) return ZP_ERR_ARG;
    const int T = 2 + (has_stage2 ? 1 : 0) + n_fri;
    if (T > MAX_TREES) return ZP_ERR_ARG;
    Sizes sz;
    sz.n_queries = n_queries; sz.has_stage2 = has_stage2 ? 1 : 0; sz.n_fri = n_fri;
    uint64_t *vb[MAX_TREES] = {}, *pb[MAX_TREES] = {};
    size_t vo = 0, po = 0;
    for (int t = 0; t < T; t++) {
        if (widths[t] < 0 || depths[t] < 0) return ZP_ERR_ARG;
        sz.w[t] = widths[t]; sz.d[t] = depths[t];
        vb[t] = values + vo; pb[t] = paths + po;
        vo += (size_t)n_queries * widths[t];
        po += (size_t)n_queries * depths[t] * 4;
    }
    Cur c{text + q_begin, text + q_end};
    if (!c.need('[')) return ZP_ERR_ARG;
    for (int q = 0; q < n_queries; q++) {
        uint64_t *v[MAX_TREES] = {}, *p[MAX_TREES] = {};
        for (int t = 0; t < T; t++) { v[t] = vb[t] + (size_t)q * sz.w[t]; p[t] = pb[t] + (size_t)q * sz.d[t] * 4; }
        if (!query(c, sz, false, index + q, v, p)) return ZP_ERR_ARG;
        if (q + 1 < n_queries && !c.need(',')) return ZP_ERR_ARG;
    }
    return c.need(']') ? ZP_OK : ZP_ERR_ARG;
}

}  // extern "C"
