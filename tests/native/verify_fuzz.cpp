// Host sanitizer driver for eigen_zeth_amd/csrc/verify.hip (built by tests/test_verify_fuzz.py with -fsanitize=address,undefined): the verifier's
// side of a constraint program.  The program blob is the engine's own, but the public inputs and the evaluations it is run on come out of a
// client's proof text (GenFinalProof, prover.proto:130-148), and a blob that reaches a C ABI must never be trusted to be well formed: the file
// named on the command line holds a valid case ([program words][n_pubchal][pubchal][logn][root32][zeta 3][ev_z][ev_zw][expected K x 3]); it is run
// as it stands (result compared), then under seeded mutations of the program, the public inputs and the evaluations -- every call must return
// ZP_OK or an error code, and no sanitizer may fire.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int32_t zp_program_eval_ext(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn, uint64_t root32,
                                       const uint64_t zeta[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, int32_t n_cols, uint64_t *h_out, int32_t n_out,
                                       int32_t threads);

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
    const size_t pw = (size_t)take(1)[0];
    const std::vector<uint64_t> prog = take(pw);
    const size_t npc = (size_t)take(1)[0];
    const std::vector<uint64_t> pubchal = take(npc);
    const int logn = (int)take(1)[0];
    const uint64_t root32 = take(1)[0];
    const std::vector<uint64_t> zeta = take(3);
    const size_t Wt = (size_t)(prog[1] + prog[2]), K = (size_t)prog[8];
    const std::vector<uint64_t> evz = take(3 * Wt), evzw = take(3 * Wt), want = take(3 * K);
    if (at != all.size()) { printf("FAIL case file\n"); return 1; }
    // exactly-sized heap buffers: the red zones sit right behind them
    auto run = [&](const std::vector<uint64_t> &p, const std::vector<uint64_t> &pc, const std::vector<uint64_t> &ez, const std::vector<uint64_t> &ezw, int lg,
                   std::vector<uint64_t> *out) {
        uint64_t *pp = (uint64_t *)malloc(p.size() * 8 + 8), *pcp = (uint64_t *)malloc(pc.size() * 8 + 8), *a = (uint64_t *)malloc(ez.size() * 8 + 8),
                 *b = (uint64_t *)malloc(ezw.size() * 8 + 8);
        memcpy(pp, p.data(), p.size() * 8); memcpy(pcp, pc.data(), pc.size() * 8); memcpy(a, ez.data(), ez.size() * 8); memcpy(b, ezw.data(), ezw.size() * 8);
        // the caller's arrays keep THEIR sizes (Wt evaluation triples, K outputs): a mutated header that names other sizes must be refused
        uint64_t *o = (uint64_t *)malloc(3 * K * 8 + 8);
        const int32_t rc = zp_program_eval_ext(pp, p.size(), pcp, (int32_t)pc.size(), lg, root32, zeta.data(), a, b, (int32_t)Wt, o, (int32_t)K, 3);
        if (out && rc == 0) out->assign(o, o + 3 * K);
        free(pp); free(pcp); free(a); free(b); free(o);
        return rc;
    };
    std::vector<uint64_t> got;
    if (run(prog, pubchal, evz, evzw, logn, &got) != 0 || got != want) { printf("FAIL the valid case\n"); return 1; }
    uint64_t s = 0x2545F4914F6CDD1DULL;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    int ok = 0, refused = 0;
    const int iters = argc > 2 ? atoi(argv[2]) : 3000;
    for (int it = 0; it < iters; it++) {
        std::vector<uint64_t> p = prog, pc = pubchal, ez = evz, ezw = evzw;
        int lg = logn;
        const int n_mut = 1 + (int)(rnd() % 3);
        for (int m = 0; m < n_mut; m++) {
            switch (rnd() % 9) {
            case 0: p[rnd() % 12] = rnd() % 5000; break;                                   // a header field
            case 1: p[rnd() % 12] = rnd(); break;
            case 2: p[12 + rnd() % (p.size() - 12)] = rnd(); break;                        // anything behind it
            case 3: p[12 + rnd() % (p.size() - 12)] ^= 1ULL << (rnd() % 64); break;
            case 4: p.resize(12 + rnd() % (p.size() - 12)); break;                         // truncated
            case 5: if (!pc.empty()) pc[rnd() % pc.size()] = rnd(); break;
            case 6: ez[rnd() % ez.size()] = rnd(); break;
            case 7: pc.resize(rnd() % (pc.size() + 2)); break;
            default: lg = (int)(rnd() % 40); break;
            }
            if (p.size() < 13) break;
        }
        const int32_t rc = run(p, pc, ez, ezw, lg, nullptr);
        if (rc == 0) ok++; else refused++;
    }
    printf("ok: %d mutated cases evaluated, %d refused\n", ok, refused);
    return 0;
}
