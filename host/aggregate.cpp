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

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int32_t rc_ = (call);                                                                   \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? zp_last_error(ctx) : "no context"); return 1; } \
    } while (0)

int main(int argc, char **argv) {
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
    const uint64_t n_proofs = desc[4], has_s2 = desc[15] != 0, pow_inner = desc[19], n_pub_inner = desc[18];
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
        Inner &I = in[p];
        I.text = read_text(files[p]);
        size_t qb, qe;
        int32_t nq, s2, nf, w[48], d[48];
        if (zp_proof_queries_scan(I.text.data(), I.text.size(), &qb, &qe, &nq, &s2, &nf, w, d, 48) != 0) { fprintf(stderr, "%s: not a proof of this prover\n", files[p]); return 2; }
        const int T = 2 + s2 + nf;
        size_t sw = 0, sd = 0;
        for (int t = 0; t < T; t++) { sw += (size_t)w[t]; sd += (size_t)d[t]; }
        if ((uint64_t)nq != desc[11] || (uint64_t)T != desc[5] || (uint64_t)s2 != has_s2) { fprintf(stderr, "%s: not the shape this directory was exported for\n", files[p]); return 2; }
        I.index.resize(nq); I.values.resize((size_t)nq * sw); I.paths.resize((size_t)nq * sd * 4);
        if (zp_proof_queries_parse(I.text.data(), qb, qe, nq, s2, nf, w, d, I.index.data(), I.values.data(), I.paths.data()) != 0) return 2;
        // the header: the text without its "queries" member (the key starts 10 bytes before the value: "queries": -- compact writers)
        size_t kb = I.text.rfind("\"queries\"", qb);
        if (kb == std::string::npos) return 2;
        size_t ke = qe;
        if (ke < I.text.size() && I.text[ke] == ',') ke++;                 // the member and the comma behind it ...
        else if (kb > 0 && I.text[kb - 1] == ',') kb--;                      // ... or, as the last member, the comma before it
        I.header = I.text.substr(0, kb) + I.text.substr(ke);
        // its statement must be the one the shape was exported for
        size_t ab, ae;
        if (zp_json_key_span(I.header.data(), I.header.size(), "air_digest", &ab, &ae) != 0 || I.header.substr(ab, ae - ab) != std::string("\"") + dg_hex + "\"") {
            fprintf(stderr, "%s: proof of another statement\n", files[p]);
            return 2;
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
        if (!ok) { fprintf(stderr, "%s: header fields missing\n", files[p]); return 2; }
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
