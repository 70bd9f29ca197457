// Host sanitizer driver for the host half of eigen_zeth_amd/csrc/r1cs.hip (built by tests/test_r1cs_fuzz.py with -DZP_R1CS_HOST_ONLY
// -fsanitize=address,undefined): the circuit-blob parser ("PZR1CS02": Poseidon template + explicit constraints + ARITHMETIC TEMPLATES with their
// witness programs, round 6), the host evaluator that runs those programs and checks every row, and the assignment script / openings record reader
// (zp_wrap_assign).  These blobs are the service's own, but whatever reaches a C ABI must not be trusted to be well formed.  The case file holds
//   [blob words][blob][n_set][wire ids][values x 4][expected public input x 4] [script words][script][openings words][openings][n_aux][aux x 4]
// The valid case is run as it stands (public input compared), then under seeded mutations; every call must return ZP_OK or an error code, and no
// sanitizer may fire.  Array sizes follow the (mutated) header, as a caller's do -- a header that names more than a small multiple of the original
// sizes is what a caller with fixed buffers would refuse itself, and is skipped.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int32_t zp_r1cs_eval(const uint64_t *circ, size_t words, uint64_t *witness, uint8_t *set, uint64_t *a_ev, uint64_t *b_ev, uint64_t *c_ev, int64_t *bad);
extern "C" int32_t zp_wrap_assign(const uint64_t *script, size_t script_words, const uint64_t *openings, size_t open_words, const uint64_t *aux, size_t n_aux,
                                  uint64_t *out_idx, uint64_t *out_val, size_t cap, size_t *n_set);

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    std::vector<uint64_t> all;
    uint64_t w;
    while (fread(&w, 8, 1, f) == 1) all.push_back(w);
    fclose(f);
    size_t at = 0;
    auto take = [&](size_t n) { const uint64_t *p = all.data() + at; at += n; return std::vector<uint64_t>(p, p + n); };
    const std::vector<uint64_t> blob = take((size_t)take(1)[0]);
    const size_t n_set = (size_t)take(1)[0];
    const std::vector<uint64_t> sidx = take(n_set), sval = take(4 * n_set), want = take(4);
    const std::vector<uint64_t> script = take((size_t)take(1)[0]), opens = take((size_t)take(1)[0]);
    const size_t n_aux = (size_t)take(1)[0];
    const std::vector<uint64_t> aux = take(4 * n_aux);
    if (at != all.size()) { printf("FAIL case file\n"); return 1; }
    const uint64_t n_wires0 = blob[1], logm0 = blob[3];

    auto eval = [&](const std::vector<uint64_t> &b, std::vector<uint64_t> *pub) -> int32_t {
        const uint64_t nw = b[1], lm = b[3];
        if (nw > 2 * n_wires0 + 64 || lm > logm0 + 1 || nw < 2) return -1000;          // a caller with buffers for THIS circuit refuses such a header itself
        const size_t m = (size_t)1 << lm;
        uint64_t *bb = (uint64_t *)malloc(b.size() * 8 + 8), *wv = (uint64_t *)calloc(nw * 4 + 1, 8), *a = (uint64_t *)malloc(m * 32 + 8), *bv = (uint64_t *)malloc(m * 32 + 8),
                 *cv = (uint64_t *)malloc(m * 32 + 8);
        uint8_t *set = (uint8_t *)calloc(nw + 1, 1);
        memcpy(bb, b.data(), b.size() * 8);
        for (size_t k = 0; k < n_set; k++)
            if (sidx[k] < nw) { memcpy(wv + 4 * sidx[k], &sval[4 * k], 32); set[sidx[k]] = 1; }
        int64_t bad = -1;
        const int32_t rc = zp_r1cs_eval(bb, b.size(), wv, set, a, bv, cv, &bad);
        if (pub && rc == 0) pub->assign(wv + 4, wv + 8);
        free(bb); free(wv); free(a); free(bv); free(cv); free(set);
        return rc;
    };
    std::vector<uint64_t> got;
    if (eval(blob, &got) != 0 || got != want) { printf("FAIL the valid case\n"); return 1; }
    auto assign = [&](const std::vector<uint64_t> &sc, const std::vector<uint64_t> &op, const std::vector<uint64_t> &ax) -> int32_t {
        const size_t cap = sc.size() > 2 && sc[2] < (1u << 22) ? (size_t)sc[2] : 0;
        uint64_t *s = (uint64_t *)malloc(sc.size() * 8 + 8), *o = (uint64_t *)malloc(op.size() * 8 + 8), *x = (uint64_t *)malloc(ax.size() * 8 + 8),
                 *oi = (uint64_t *)malloc(cap * 8 + 8), *ov = (uint64_t *)malloc(cap * 32 + 8);
        memcpy(s, sc.data(), sc.size() * 8); memcpy(o, op.data(), op.size() * 8); memcpy(x, ax.data(), ax.size() * 8);
        size_t n = 0;
        const int32_t rc = zp_wrap_assign(s, sc.size(), o, op.size(), x, ax.size() / 4, oi, ov, cap, &n);
        free(s); free(o); free(x); free(oi); free(ov);
        return rc;
    };
    if (assign(script, opens, aux) != 0) { printf("FAIL the valid assignment\n"); return 1; }

    uint64_t s = 0x9E3779B97F4A7C15ULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    auto mutate = [&](std::vector<uint64_t> &v, size_t hot_lo, size_t hot_hi) {
        const int n_mut = 1 + (int)(rnd() % 3);
        for (int k = 0; k < n_mut; k++) {
            size_t pos = rnd() % v.size();
            if (rnd() % 2 && hot_hi > hot_lo) pos = hot_lo + rnd() % (hot_hi - hot_lo);        // headers and section tables: where the indices live
            switch (rnd() % 6) {
                case 0: v[pos] ^= 1ULL << (rnd() % 64); break;
                case 1: v[pos] = rnd() % 64; break;
                case 2: v[pos] = ~0ULL - (rnd() % 4); break;
                case 3: v[pos] += 1; break;
                case 4: v[pos] -= 1; break;
                default: v[pos] = rnd(); break;
            }
        }
        if (rnd() % 16 == 0 && v.size() > 4) v.resize(v.size() - 1 - rnd() % 3);                 // truncated
    };
    // where the arithmetic templates start: behind the Poseidon template, instances, waves and the explicit constraints -- found by the first header
    // whose nine counts are followed by three zeros and whose first_row is plausible; simpler and robust enough: the last third of the blob
    const size_t arith_lo = blob.size() * 2 / 3;
    const int iters = argc > 2 ? atoi(argv[2]) : 1500;
    int ok = 0, refused = 0, skipped = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint64_t> b = blob;
        if (it % 3 == 0) mutate(b, 0, 16);
        else mutate(b, arith_lo, b.size());
        const int32_t rc = eval(b, nullptr);
        if (rc == 0) ok++; else if (rc == -1000) skipped++; else refused++;
        std::vector<uint64_t> sc = script, op = opens, ax = aux;
        switch (it % 3) {
            case 0: mutate(sc, 0, 8 + 3 * 8); break;
            case 1: mutate(op, 0, 32); break;
            default: mutate(op, op.size() - 64, op.size()); break;
        }
        (void)assign(sc, op, ax);
    }
    printf("ok: valid case matches; %d mutated blobs evaluated, %d refused, %d skipped by the caller's own size check\n", ok, refused, skipped);
    return 0;
}
