// Host sanitizer driver for eigen_zeth_amd/csrc/proofparse.hip (built by tests/test_proofparse_fuzz.py with -fsanitize=address,undefined).
// The text of a recursive proof comes from the client (proto/prover/v1/prover.proto:115-148), so the parser must refuse, not corrupt:
// hand-made hostile documents (repeated keys, extra FRI layers behind a valid one, sizes that disagree with the scan) and seeded
// mutations of a well-formed document.  Exit code 0 + "ok" when every call returned ZP_OK or ZP_ERR_ARG and no sanitizer fired; scan and
// parse must agree (a text the scan accepts and the write pass refuses is counted, and must not write out of bounds either way).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

extern "C" {
int32_t zp_json_key_span(const char *text, size_t len, const char *key, size_t *begin, size_t *end);
int32_t zp_proof_queries_scan(const char *text, size_t len, size_t *q_begin, size_t *q_end, int32_t *n_queries, int32_t *has_stage2, int32_t *n_fri,
                              int32_t *widths, int32_t *depths, int32_t max_trees);
int32_t zp_proof_queries_parse(const char *text, size_t q_begin, size_t q_end, int32_t n_queries, int32_t has_stage2, int32_t n_fri,
                               const int32_t *widths, const int32_t *depths, uint64_t *index, uint64_t *values, uint64_t *paths);
}

static std::string opening(int w, int d, uint64_t seed) {
    std::string s = "{\"values\":[";
    for (int i = 0; i < w; i++) s += (i ? "," : "") + std::to_string(seed * 31 + i);
    s += "],\"path\":[";
    for (int i = 0; i < d; i++) {
        s += i ? ",[" : "[";
        for (int k = 0; k < 4; k++) s += (k ? "," : "") + std::to_string(seed * 7 + 4 * i + k);
        s += "]";
    }
    return s + "]}";
}

static std::string query(int idx, bool stage2, int n_fri, int extra_fri = 0, bool dup_fri = false) {
    std::string s = "{\"index\":" + std::to_string(idx) + ",\"trace\":" + opening(5, 3, idx + 1);
    if (stage2) s += ",\"stage2\":" + opening(2, 3, idx + 2);
    s += ",\"quotient\":" + opening(3, 3, idx + 3) + ",\"fri\":[";
    for (int l = 0; l < n_fri + extra_fri; l++) s += (l ? "," : "") + opening(6, 2, idx + 4 + l);
    s += "]";
    if (dup_fri) {                       // the advisor's case: a long "fri" followed by a valid one (the scan used to keep the last only)
        s += ",\"fri\":[";
        for (int l = 0; l < n_fri; l++) s += (l ? "," : "") + opening(6, 2, idx + 4 + l);
        s += "]";
    }
    return s + "}";
}

static std::string doc(const std::vector<std::string> &qs) {
    std::string s = "{\"params\":{\"logn\":3},\"queries\":[";
    for (size_t i = 0; i < qs.size(); i++) s += (i ? "," : "") + qs[i];
    return s + "],\"tail\":[1,2,{\"queries\":0}]}";
}

// scan + parse into exactly-sized heap buffers (ASan's red zones sit right behind them).  Returns 0 when refused, 1 when accepted.
static int run(const std::string &text, int *disagree) {
    size_t qb = 0, qe = 0;
    int32_t nq = 0, s2 = 0, nf = 0, w[48], d[48];
    int32_t rc = zp_proof_queries_scan(text.data(), text.size(), &qb, &qe, &nq, &s2, &nf, w, d, 48);
    if (rc != 0) return 0;
    const int T = 2 + s2 + nf;
    size_t nv = 0, np = 0;
    for (int t = 0; t < T; t++) { nv += (size_t)nq * w[t]; np += (size_t)nq * d[t] * 4; }
    // dirty stack below the call, so that an uninitialised pointer would be a recognisable wild address
    volatile uint64_t dirt[512];
    for (int i = 0; i < 512; i++) dirt[i] = 0x4141414141414140ULL;
    (void)dirt;
    uint64_t *index = (uint64_t *)malloc(nq * 8 + 8), *values = (uint64_t *)malloc(nv * 8 + 8), *paths = (uint64_t *)malloc(np * 8 + 8);
    rc = zp_proof_queries_parse(text.data(), qb, qe, nq, s2, nf, w, d, index, values, paths);
    free(index); free(values); free(paths);
    if (rc != 0) { (*disagree)++; return 0; }
    return 1;
}

int main() {
    int disagree = 0, accepted = 0, total = 0;
    auto expect = [&](const std::string &t, int want, const char *what) {
        const int got = run(t, &disagree);
        total++;
        accepted += got;
        if (got != want) { printf("FAIL %s: accepted=%d want=%d\n", what, got, want); exit(1); }
    };
    expect(doc({query(1, false, 2), query(2, false, 2)}), 1, "plain");
    expect(doc({query(1, true, 3), query(2, true, 3), query(9, true, 3)}), 1, "stage2");
    expect(doc({query(1, false, 1, 30, true)}), 0, "repeated fri, long first");
    expect(doc({query(1, false, 1), query(2, false, 1, 30, true)}), 0, "repeated fri in a later query");
    expect(doc({query(1, false, 2), query(2, false, 2, 1)}), 0, "extra fri layer in a later query");
    expect(doc({query(1, true, 45)}), 1, "48 trees (the limit)");
    expect(doc({query(1, true, 46)}), 0, "49 trees");
    expect(doc({query(1, false, 45)}), 1, "47 trees without stage2 (slot 1 stays free while sizing)");
    expect(doc({query(1, false, 46)}), 0, "one more");
    {   // repeated keys of every kind
        std::string q = query(1, false, 1);
        for (const char *k : {"\"index\":7,", "\"trace\":{\"values\":[],\"path\":[]},", "\"quotient\":{\"values\":[1],\"path\":[]},"}) {
            std::string t = q;
            t.insert(1, k);
            expect(doc({t}), 0, k);
        }
        std::string t = doc({q});
        size_t at = t.find("\"values\":[");
        t.insert(at, "\"values\":[1,2,3,4,5,6,7,8,9],");
        expect(t, 0, "repeated values");
        t = doc({q});
        at = t.find("\"path\":[");
        t.insert(at, "\"path\":[[1,2,3,4],[1,2,3,4],[1,2,3,4],[1,2,3,4],[1,2,3,4]],");
        expect(t, 0, "repeated path");
    }
    {   // the write pass against sizes that are not the text's (a caller error must be refused, not written through)
        const std::string t = doc({query(1, false, 2), query(2, false, 2)});
        size_t qb = 0, qe = 0;
        int32_t nq = 0, s2 = 0, nf = 0, w[48], d[48];
        if (zp_proof_queries_scan(t.data(), t.size(), &qb, &qe, &nq, &s2, &nf, w, d, 48) != 0) { printf("FAIL scan\n"); return 1; }
        std::vector<uint64_t> idx(nq), v(4096), p(4096);
        for (int mode = 0; mode < 4; mode++) {
            int32_t w2[48], d2[48];
            memcpy(w2, w, sizeof w); memcpy(d2, d, sizeof d);
            int32_t nf2 = nf, nq2 = nq;
            if (mode == 0) nf2 = nf - 1;
            if (mode == 1) w2[0]--;
            if (mode == 2) d2[1]--;
            if (mode == 3) nq2 = nq + 1;
            std::vector<uint64_t> idx2(nq2);
            if (zp_proof_queries_parse(t.data(), qb, qe, nq2, s2, nf2, w2, d2, idx2.data(), v.data(), p.data()) == 0) {
                printf("FAIL mismatched sizes accepted (mode %d)\n", mode);
                return 1;
            }
        }
    }
    // seeded mutations: byte flips, deletions, duplications of slices, truncations
    const std::string base = doc({query(3, true, 2), query(4, true, 2), query(5, true, 2)});
    uint64_t s = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    const char alphabet[] = "{}[],:\"0123456789 -.e\\afirtq";
    for (int it = 0; it < 20000; it++) {
        std::string t = base;
        const int n_mut = 1 + (int)(rnd() % 3);
        for (int m = 0; m < n_mut && !t.empty(); m++) {
            const size_t at = rnd() % t.size();
            switch (rnd() % 5) {
            case 0: t[at] = alphabet[rnd() % (sizeof alphabet - 1)]; break;
            case 1: t.erase(at, 1 + rnd() % 8); break;
            case 2: { const size_t len = 1 + rnd() % 200; t.insert(at, t.substr(at, len)); break; }
            case 3: t.resize(at); break;
            default: { const size_t from = rnd() % t.size(); t.insert(at, t.substr(from, 1 + rnd() % 120)); break; }
            }
        }
        accepted += run(t, &disagree);
        total++;
        size_t b, e;
        zp_json_key_span(t.data(), t.size(), "queries", &b, &e);
    }
    printf("ok: %d texts, %d accepted, %d accepted by the scan and refused by the write pass\n", total, accepted, disagree);
    return 0;
}
