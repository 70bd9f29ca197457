// GenAggregatedProof from a compiled host on the C-ABI alone -- what a Rust prover service does with the two proof strings of the request
// (proto/prover/v1/prover.proto:115-126; the client side is src/prover/provider.rs:422-451), written in C++ because this image has no Rust
// toolchain.  No Python, no torch, no compiler at run time: the verifier AIR and its witness schedule are data files
// (tools/export_recursion_shape.py writes them once per shape of inner proofs).
//
// usage: aggregate <shape dir> <batch id> <proof1.json> <proof2.json> <out.json>
//   parse the openings of both proof texts (zp_proof_queries_scan / _parse), read the few header fields the transcripts absorb, build the
//   whole verifier witness in HBM (zp_recursion_witness), prove it (zp_stark_prove), write the aggregated proof: the two HEADERS (the proof
//   texts without their "queries") + the STARK.  The output is byte for byte what the Python service's engine answers for the same request
//   (tests/test_gpu_native_prover.py).  With one chunk the client sends the same proof twice (provider.rs:386-387): pass it twice, it is
//   verified once -- the shape directory must then have been exported with n_proofs = 1.
//
// usage: aggregate final <final dir> <aggregated proof.json> <aggregator address> <blinding seed> <out proof.json> <out public_input.json>
//   GenFinalProof (prover.proto:130-148; the client side is src/prover/provider.rs:472-503): the aggregation STARK inside the aggregated proof is
//   the one inner proof of a verifier AIR one level up; its witness (zp_recursion_witness) is proven in BN128-hash mode (zp_stark_prove_bn128), the
//   prover's binary openings (zp_stark_openings) drive the wrap circuit's assignment (zp_wrap_assign) and zp_groth16_prove makes the proof under
//   the key files of the directory (tools/export_recursion_shape.py: export_final).  With the service's deterministic blinding (EngineConfig.
//   groth16_seed = the seed given here) the two output files are byte for byte the engine's answer (tests/test_gpu_native_prover.py).
// build: make -C host
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/zeth_prover.h"

static std::string read_text(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    std::string s;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}

static std::vector<uint64_t> read_words(const std::string &path) {
    const std::string s = read_text(path.c_str());
    if (s.size() % 8) { fprintf(stderr, "bad file %s\n", path.c_str()); exit(2); }
    std::vector<uint64_t> v(s.size() / 8);
    memcpy(v.data(), s.data(), s.size());
    return v;
}

// the numbers of the (possibly nested) JSON array that is the value of member `path[0]`.`path[1]`... of the object text[0 .. len), appended
// to out in reading order.  Only what the provers write: non-negative integers.
static bool numbers_at(const char *text, size_t len, const char *const *path, int depth, std::vector<uint64_t> *out) {
    size_t b = 0, e = 0;
    if (zp_json_key_span(text, len, path[0], &b, &e) != 0) return false;
    if (depth > 1) return numbers_at(text + b, e - b, path + 1, depth - 1, out);
    for (size_t i = b; i < e;) {
        const char c = text[i];
        if (c >= '0' && c <= '9') {
            uint64_t v = 0;
            while (i < e && text[i] >= '0' && text[i] <= '9') { v = v * 10 + (uint64_t)(text[i] - '0'); i++; }
            out->push_back(v);
        } else if (c == '[' || c == ']' || c == ',' || c == ' ' || c == '\n') {
            i++;
        } else {
            return false;
        }
    }
    return true;
}

struct Inner {
    std::string text, header;
    std::vector<uint64_t> index, values, paths, stream;
};

// openings, header (the text without its "queries") and transcript stream of one inner proof text; false (with a message) when it is not a
// proof of the statement / shape the descriptor was exported for
static bool parse_inner(Inner &I, const char *name, const std::vector<uint64_t> &desc, const char *dg_hex, const uint64_t *dgw) {
    const uint64_t has_s2 = desc[15] != 0, pow_inner = desc[19], n_pub_inner = desc[18];
    size_t qb, qe;
    int32_t nq, s2, nf, w[48], d[48];
    if (zp_proof_queries_scan(I.text.data(), I.text.size(), &qb, &qe, &nq, &s2, &nf, w, d, 48) != 0) { fprintf(stderr, "%s: not a proof of this prover\n", name); return false; }
    const int T = 2 + s2 + nf;
    size_t sw = 0, sd = 0;
    for (int t = 0; t < T; t++) { sw += (size_t)w[t]; sd += (size_t)d[t]; }
    if ((uint64_t)nq != desc[11] || (uint64_t)T != desc[5] || (uint64_t)s2 != has_s2) { fprintf(stderr, "%s: not the shape this directory was exported for\n", name); return false; }
    I.index.resize(nq); I.values.resize((size_t)nq * sw); I.paths.resize((size_t)nq * sd * 4);
    if (zp_proof_queries_parse(I.text.data(), qb, qe, nq, s2, nf, w, d, I.index.data(), I.values.data(), I.paths.data()) != 0) return false;
    // the header: the text without its "queries" member (the key starts 10 bytes before the value: "queries": -- compact writers)
    size_t kb = I.text.rfind("\"queries\"", qb);
    if (kb == std::string::npos) return false;
    size_t ke = qe;
    if (ke < I.text.size() && I.text[ke] == ',') ke++;                 // the member and the comma behind it ...
    else if (kb > 0 && I.text[kb - 1] == ',') kb--;                      // ... or, as the last member, the comma before it
    I.header = I.text.substr(0, kb) + I.text.substr(ke);
    // its statement must be the one the shape was exported for
    size_t ab, ae;
    if (zp_json_key_span(I.header.data(), I.header.size(), "air_digest", &ab, &ae) != 0 || I.header.substr(ab, ae - ab) != std::string("\"") + dg_hex + "\"") {
        fprintf(stderr, "%s: proof of another statement\n", name);
        return false;
    }
    // what the transcript absorbs: digest words | publics | roots | evaluations | FRI roots | final layer | nonce
    I.stream.assign(dgw, dgw + 4);
    const char *pubs[] = {"publics"}, *rt[] = {"roots", "trace"}, *r2[] = {"roots", "stage2"}, *rq[] = {"roots", "quotient"}, *ez[] = {"evals", "z"},
               *ew[] = {"evals", "zw"}, *fr[] = {"fri", "roots"}, *ff[] = {"fri", "final"}, *pn[] = {"pow_nonce"};
    const char *h = I.header.data();
    const size_t hl = I.header.size();
    bool ok = numbers_at(h, hl, pubs, 1, &I.stream) && I.stream.size() == 4 + n_pub_inner && numbers_at(h, hl, rt, 2, &I.stream);
    if (ok && has_s2) ok = numbers_at(h, hl, r2, 2, &I.stream);
    ok = ok && numbers_at(h, hl, rq, 2, &I.stream) && numbers_at(h, hl, ez, 2, &I.stream) && numbers_at(h, hl, ew, 2, &I.stream) &&
         numbers_at(h, hl, fr, 2, &I.stream) && numbers_at(h, hl, ff, 2, &I.stream);
    if (ok && pow_inner) ok = numbers_at(h, hl, pn, 1, &I.stream);
    if (!ok) { fprintf(stderr, "%s: header fields missing\n", name); return false; }
    return true;
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int32_t rc_ = (call);                                                                   \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? zp_last_error(ctx) : "no context"); return 1; } \
    } while (0)

// ---- GenFinalProof
typedef unsigned __int128 u128;
static const uint64_t FR_MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static bool geq_mod(const uint64_t *a) {
    for (int i = 3; i >= 0; i--) { if (a[i] != FR_MOD[i]) return a[i] > FR_MOD[i]; }
    return true;
}
static void reduce_mod(uint64_t *a) {          // a < 2^256 -> a mod r (at most five subtractions)
    while (geq_mod(a)) {
        u128 b = 0;
        for (int i = 0; i < 4; i++) { const u128 d = (u128)a[i] - FR_MOD[i] - (uint64_t)b; a[i] = (uint64_t)d; b = (d >> 64) & 1; }
    }
}
static bool dec_to_words(const std::string &s, uint64_t *w) {   // a decimal integer < 2^256
    w[0] = w[1] = w[2] = w[3] = 0;
    if (s.empty()) return false;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        u128 carry = (u128)(c - '0');
        for (int i = 0; i < 4; i++) { carry += (u128)w[i] * 10; w[i] = (uint64_t)carry; carry >>= 64; }
        if (carry) return false;
    }
    return true;
}
static std::string words_to_dec(const uint64_t *w4) {
    uint64_t v[4] = {w4[0], w4[1], w4[2], w4[3]};
    std::string d;
    while (v[0] | v[1] | v[2] | v[3]) {
        u128 rem = 0;
        for (int k = 3; k >= 0; k--) { const u128 cur = (rem << 64) | v[k]; v[k] = (uint64_t)(cur / 10); rem = cur % 10; }
        d.insert(d.begin(), (char)('0' + (int)rem));
    }
    return d.empty() ? "0" : d;
}
static void digest_words(const std::string &s, uint64_t *w) {    // SHA-256 read as a big-endian integer, mod r
    uint8_t dg[32];
    zp_sha256((const uint8_t *)s.data(), s.size(), dg);
    for (int i = 0; i < 4; i++) { w[i] = 0; for (int b = 0; b < 8; b++) w[i] = (w[i] << 8) | dg[8 * (3 - i) + b]; }
    reduce_mod(w);
}
static std::string limbs_dec(const uint32_t *l8) {
    uint64_t w[4];
    for (int i = 0; i < 4; i++) w[i] = (uint64_t)l8[2 * i] | ((uint64_t)l8[2 * i + 1] << 32);
    return words_to_dec(w);
}

static int run_final(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: aggregate final <final dir> <aggregated proof.json> <aggregator address> <blinding seed> <out proof.json> <out public_input.json>\n"); return 2; }
    const std::string dir = argv[2], addr = argv[4], seed = argv[5];
    const std::vector<uint64_t> inner_prog = read_words(dir + "/inner_program.bin"), vprog = read_words(dir + "/verifier_program.bin"),
                                desc = read_words(dir + "/witness_desc.bin"), tables = read_words(dir + "/poseidon_bn254_t17.bin"),
                                circ = read_words(dir + "/wrap_circuit.bin"), script = read_words(dir + "/wrap_script.bin");
    const std::string meta = read_text((dir + "/final.txt").c_str());
    int logn, logb, fri_logf, fri_final_log, n_queries;
    if (sscanf(meta.c_str(), "%d %d %d %d %d", &logn, &logb, &fri_logf, &fri_final_log, &n_queries) != 5 || desc.size() < 26 || desc[4] != 1 || tables.size() < 2 ||
        circ.size() < 16 || script.size() < 6) { fprintf(stderr, "bad final shape files\n"); return 2; }
    std::string circuit_text = meta.substr(meta.find('\n') + 1);
    while (!circuit_text.empty() && (circuit_text.back() == '\n' || circuit_text.back() == '\r')) circuit_text.pop_back();
    uint8_t dg[32];
    uint64_t dgw[4];
    if (zp_program_digest(inner_prog.data(), inner_prog.size(), dg, dgw) != 0) return 2;
    char dg_hex[17];
    for (int i = 0; i < 8; i++) snprintf(dg_hex + 2 * i, 3, "%02x", dg[i]);
    // the one inner proof: the aggregation STARK of the aggregated proof
    const std::string agg = read_text(argv[3]);
    size_t sb, se, kb2, ke2;
    if (zp_json_key_span(agg.data(), agg.size(), "kind", &kb2, &ke2) != 0 || agg.substr(kb2, ke2 - kb2) != "\"aggregated\"" ||
        zp_json_key_span(agg.data(), agg.size(), "stark", &sb, &se) != 0) { fprintf(stderr, "%s: not an aggregated proof of this prover\n", argv[3]); return 2; }
    Inner I;
    I.text = agg.substr(sb, se - sb);
    if (!parse_inner(I, argv[3], desc, dg_hex, dgw)) return 2;
    zp_ctx *ctx = nullptr;
    CHECK(zp_create(&ctx, 0));
    const uint64_t rp = tables[0];
    if (tables.size() != 1 + ((8 + rp) * 17 + 17 * 17) * 4) { fprintf(stderr, "bad Poseidon-BN254 table file\n"); return 2; }
    CHECK(zp_set_poseidon_bn254(ctx, 17, (int32_t)rp, tables.data() + 1, tables.data() + 1 + (8 + rp) * 17 * 4));
    const size_t npub = zp_recursion_publics_words(desc.data(), desc.size());
    const size_t N = (size_t)32 * desc[1] * desc[2];
    if (npub == 0 || N != ((size_t)1 << logn)) { fprintf(stderr, "descriptor and parameters disagree\n"); return 2; }
    std::vector<uint64_t> pubs(npub);
    void *d_trace = nullptr;
    CHECK(zp_dev_alloc(ctx, 47 * N * 8, &d_trace));
    const uint64_t *pi = I.index.data(), *pv = I.values.data(), *pp = I.paths.data(), *ps = I.stream.data();
    const size_t sw = I.stream.size();
    CHECK(zp_recursion_witness(ctx, desc.data(), desc.size(), &pi, &pv, &pp, &ps, &sw, (uint64_t *)d_trace, pubs.data(), npub, 0));
    char *fs = nullptr;
    size_t fs_len = 0;
    CHECK(zp_stark_prove_bn128(ctx, "mverify", vprog.data(), vprog.size(), (const uint64_t *)d_trace, 47 * N, pubs.data(), (int32_t)npub, logn, logb, fri_logf,
                               fri_final_log, n_queries, &fs, &fs_len));
    uint8_t fsd[32];
    zp_sha256((const uint8_t *)fs, fs_len, fsd);
    char fs_hex[65];
    for (int i = 0; i < 32; i++) snprintf(fs_hex + 2 * i, 3, "%02x", fsd[i]);
    // the wrap: the caller-set wires from the prover's own openings record, bound to the aggregator address
    const uint64_t *op = nullptr;
    size_t op_words = 0;
    CHECK(zp_stark_openings(ctx, &op, &op_words));
    uint64_t addr4[4];
    if (dec_to_words(addr, addr4)) reduce_mod(addr4); else digest_words(addr, addr4);
    // stage B-2: the circuit takes the statement's sparse fixed columns at zeta from its caller (a function of the statement, the public inputs and zeta)
    std::vector<uint64_t> aux(4 * (size_t)vprog[3]);
    size_t n_aux = 0;
    uint64_t zeta3[3], root32 = 0;
    CHECK(zp_get_constants(ctx, ZP_CONST_ROOT32, &root32, 1));
    CHECK(zp_wrap_aux(op, op_words, vprog.data(), vprog.size(), pubs.data(), (int32_t)npub, logn, root32, addr4, aux.data(), (size_t)vprog[3], &n_aux, zeta3));
    std::vector<uint64_t> set_idx(script[2]), set_val(4 * script[2]);
    size_t n_set = 0;
    CHECK(zp_wrap_assign(script.data(), script.size(), op, op_words, aux.data(), n_aux, set_idx.data(), set_val.data(), set_idx.size(), &n_set));
    // deterministic blinding, the engine's rule (EngineConfig.groth16_seed): SHA-256(seed|final STARK digest|address|tag) mod r, or 1
    uint64_t r[4], sc[4];
    for (int k = 0; k < 2; k++) {
        uint64_t *dst = k ? sc : r;
        digest_words(seed + "|" + fs_hex + "|" + addr + "|" + (k ? "s" : "r"), dst);
        if (!(dst[0] | dst[1] | dst[2] | dst[3])) dst[0] = 1;
    }
    // the key's points: files -> HBM
    const size_t n_wires = circ[1], m = (size_t)1 << circ[3];
    const std::string vw = read_text((dir + "/key_v_wires.bin").c_str());          // the wires with a non-zero column in B
    const size_t n_v = vw.size() / 4;
    if (vw.size() % 4 || n_v < 1 || n_v > n_wires) { fprintf(stderr, "bad key_v_wires.bin\n"); return 2; }
    struct KeyPart { const char *file; size_t bytes; void *dev; } parts[6] = {{"/key_u1x.bin", (n_wires + 2) * 64, nullptr}, {"/key_v1x.bin", (n_v + 2) * 64, nullptr},
                                                                               {"/key_v2x.bin", (n_v + 2) * 128, nullptr}, {"/key_l1.bin", n_wires * 64, nullptr},
                                                                               {"/key_h1.bin", (m - 1) * 64, nullptr}, {"/key_v_wires.bin", n_v * 4, nullptr}};
    for (KeyPart &kp : parts) {
        const std::string raw = read_text((dir + kp.file).c_str());
        if (raw.size() != kp.bytes) { fprintf(stderr, "%s: %zu bytes, expected %zu\n", kp.file, raw.size(), kp.bytes); return 2; }
        CHECK(zp_dev_alloc(ctx, kp.bytes, &kp.dev));
        CHECK(zp_h2d(ctx, kp.dev, raw.data(), kp.bytes));
    }
    const std::string d1 = read_text((dir + "/key_delta1.bin").c_str());
    if (d1.size() != 64) { fprintf(stderr, "bad key_delta1.bin\n"); return 2; }
    uint32_t pa[16], pb[32], pc[16];
    std::vector<uint64_t> pub(4 * circ[9]);
    double ms[8];
    int64_t bad = -1;
    CHECK(zp_groth16_prove(ctx, circ.data(), circ.size(), (const uint32_t *)parts[0].dev, (const uint32_t *)parts[5].dev, n_v, (const uint32_t *)parts[1].dev, (const uint32_t *)parts[2].dev,
                           (const uint32_t *)parts[3].dev, (const uint32_t *)parts[4].dev, (const uint32_t *)d1.data(), set_idx.data(), set_val.data(), n_set, r, sc,
                           pa, pb, pc, pub.data(), ms, &bad));
    // the text eigen-zeth parses (src/settlement/ethereum/mod.rs:445-481), written as json.dumps writes it
    std::string js = "{\"pi_a\": {\"x\": \"" + limbs_dec(pa) + "\", \"y\": \"" + limbs_dec(pa + 8) + "\"}, \"pi_b\": {\"x\": [\"" + limbs_dec(pb) + "\", \"" + limbs_dec(pb + 8) +
                     "\"], \"y\": [\"" + limbs_dec(pb + 16) + "\", \"" + limbs_dec(pb + 24) + "\"]}, \"pi_c\": {\"x\": \"" + limbs_dec(pc) + "\", \"y\": \"" + limbs_dec(pc + 8) +
                     "\"}, \"protocol\": \"groth16\", \"curve\": \"BN128\", \"circuit\": \"" + circuit_text + "\", \"final_stark_sha256\": \"" + fs_hex + "\", \"zeta\": [\"" +
                     std::to_string(zeta3[0]) + "\", \"" + std::to_string(zeta3[1]) + "\", \"" + std::to_string(zeta3[2]) + "\"]}";
    const std::string pj = "[\"" + words_to_dec(pub.data()) + "\"]";
    for (int k = 0; k < 2; k++) {
        const std::string &out = k ? pj : js;
        FILE *o = fopen(argv[6 + k], "wb");
        if (!o || fwrite(out.data(), 1, out.size(), o) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[6 + k]); return 2; }
        fclose(o);
    }
    printf("final proof: final STARK %zu bytes (sha256 %.16s...), wrap %zu wires set, witness %.1f ms + QAP %.1f ms + MSMs %.1f ms -> %s\n", fs_len, fs_hex, n_set, ms[0], ms[1],
           ms[2], argv[6]);
    zp_free_buffer(fs);
    for (KeyPart &kp : parts) zp_dev_free(ctx, kp.dev);
    zp_dev_free(ctx, d_trace);
    zp_destroy(ctx);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && !strcmp(argv[1], "final")) return run_final(argc, argv);
    if (argc < 6) { fprintf(stderr, "usage: aggregate <shape dir> <batch id> <proof1.json> <proof2.json> <out.json>\n"); return 2; }
    const std::string dir = argv[1];
    const std::vector<uint64_t> inner_prog = read_words(dir + "/inner_program.bin"), vprog = read_words(dir + "/verifier_program.bin"),
                                desc = read_words(dir + "/witness_desc.bin");
    const std::string meta = read_text((dir + "/aggregate.txt").c_str());
    int logn, logb, fri_logf, fri_final_log, n_queries, pow_bits;
    if (sscanf(meta.c_str(), "%d %d %d %d %d %d", &logn, &logb, &fri_logf, &fri_final_log, &n_queries, &pow_bits) != 6) { fprintf(stderr, "bad aggregate.txt\n"); return 2; }
    std::string head = meta.substr(meta.find('\n') + 1);
    while (!head.empty() && (head.back() == '\n' || head.back() == '\r')) head.pop_back();
    const size_t ph = head.find("%s");
    if (ph == std::string::npos || desc.size() < 26) { fprintf(stderr, "bad shape files\n"); return 2; }
    for (const char *c = argv[2]; *c; c++)
        if (!((*c >= 'a' && *c <= 'z') || (*c >= 'A' && *c <= 'Z') || (*c >= '0' && *c <= '9') || *c == '-' || *c == '_')) { fprintf(stderr, "batch id: [A-Za-z0-9_-] only\n"); return 2; }
    head.replace(ph, 2, argv[2]);
    const uint64_t n_proofs = desc[4];
    uint8_t dg[32];
    uint64_t dgw[4];
    if (zp_program_digest(inner_prog.data(), inner_prog.size(), dg, dgw) != 0) return 2;
    char dg_hex[17];
    for (int i = 0; i < 8; i++) snprintf(dg_hex + 2 * i, 3, "%02x", dg[i]);
    // the request names two proofs; equal texts are verified once
    std::vector<const char *> files = {argv[3]};
    if (n_proofs == 2) files.push_back(argv[4]);
    else if (read_text(argv[3]) != read_text(argv[4])) { fprintf(stderr, "this shape directory aggregates ONE proof (the same text twice)\n"); return 2; }
    std::vector<Inner> in(files.size());
    for (size_t p = 0; p < in.size(); p++) {
        in[p].text = read_text(files[p]);
        if (!parse_inner(in[p], files[p], desc, dg_hex, dgw)) return 2;
    }
    zp_ctx *ctx = nullptr;
    CHECK(zp_create(&ctx, 0));
    const size_t npub = zp_recursion_publics_words(desc.data(), desc.size());
    const size_t N = (size_t)32 * desc[1] * desc[2];
    if (npub == 0 || N != ((size_t)1 << logn)) { fprintf(stderr, "descriptor and parameters disagree\n"); return 2; }
    std::vector<uint64_t> pubs(npub);
    void *d_trace = nullptr;
    CHECK(zp_dev_alloc(ctx, 47 * N * 8, &d_trace));
    std::vector<const uint64_t *> pi, pv, pp, ps;
    std::vector<size_t> sw;
    for (const Inner &I : in) { pi.push_back(I.index.data()); pv.push_back(I.values.data()); pp.push_back(I.paths.data()); ps.push_back(I.stream.data()); sw.push_back(I.stream.size()); }
    CHECK(zp_recursion_witness(ctx, desc.data(), desc.size(), pi.data(), pv.data(), pp.data(), ps.data(), sw.data(), (uint64_t *)d_trace, pubs.data(), npub, 0));
    char *json = nullptr;
    size_t len = 0;
    CHECK(zp_stark_prove(ctx, "mverify", vprog.data(), vprog.size(), (const uint64_t *)d_trace, 47 * N, pubs.data(), (int32_t)npub, logn, logb, fri_logf, fri_final_log,
                         n_queries, pow_bits, &json, &len));
    std::string out = head + ",\"inner\":[";
    for (size_t p = 0; p < in.size(); p++) out += (p ? "," : "") + in[p].header;
    out += "],\"stark\":";
    out.append(json, len);
    out += "}";
    FILE *o = fopen(argv[5], "wb");
    if (!o || fwrite(out.data(), 1, out.size(), o) != out.size()) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    fclose(o);
    zp_free_buffer(json);
    zp_dev_free(ctx, d_trace);
    zp_destroy(ctx);
    printf("aggregated proof: %zu bytes (%zu inner headers of %zu + %zu bytes, STARK %zu bytes) -> %s\n", out.size(), in.size(), in[0].header.size(),
           in.back().header.size(), len, argv[5]);
    return 0;
}
