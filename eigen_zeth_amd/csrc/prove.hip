// zp_stark_prove: the whole chunk-STARK prover behind ONE C-ABI call -- trace in HBM in, proof bytes out.
// Serves GenChunkProof (proto/prover/v1/prover.proto:56-66; client src/prover/provider.rs:358-390): a host in any
// language binds this entry point and needs neither the Python orchestration (eigen_zeth_amd/stark/prover.py, which stays
// the readable statement of the protocol and the harness of the parity tests) nor a compiler -- the statement is the
// constraint program blob.  Host C++ only: every O(trace) step is one of the library's own entry points (zp_lde,
// zp_merkle_commit, zp_eval_quotient, zp_poly_eval_ext, zp_deep_quotient, zp_fri_fold, ...), the Fiat-Shamir transcript
// runs through zp_poseidon_sponge.  The proof text is byte-identical to proof_to_json(prove(...)) of the Python
// orchestration on the same inputs (tests/test_gpu_native_prover.py), Goldilocks-hash mode.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace {

// ---- SHA-256 (FIPS 180-4) of the program blob: the AIR digest bound into the transcript
struct Sha256 {
    uint32_t h[8];
    static uint32_t ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
    static void block(uint32_t *h, const uint8_t *p) {
        static const uint32_t K[64] = {
            0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
            0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
            0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
            0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
            0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
            0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
        uint32_t w[64];
        for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; i++) {
            const uint32_t s0 = ror(w[i - 15], 7) ^ ror(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ror(w[i - 2], 17) ^ ror(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        for (int i = 0; i < 64; i++) {
            const uint32_t t1 = hh + (ror(e, 6) ^ ror(e, 11) ^ ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
            const uint32_t t2 = (ror(a, 2) ^ ror(a, 13) ^ ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    static void digest(const uint8_t *data, size_t len, uint8_t out[32]) {
        uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
        size_t i = 0;
        for (; i + 64 <= len; i += 64) block(h, data + i);
        uint8_t tail[128];
        const size_t rem = len - i;
        memcpy(tail, data + i, rem);
        tail[rem] = 0x80;
        const size_t padded = rem + 1 + 8 <= 64 ? 64 : 128;
        memset(tail + rem + 1, 0, padded - rem - 1);
        const uint64_t bits = (uint64_t)len * 8;
        for (int k = 0; k < 8; k++) tail[padded - 1 - k] = (uint8_t)(bits >> (8 * k));
        block(h, tail);
        if (padded == 128) block(h, tail + 64);
        for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
    }
};

// The program of a verifier AIR is megabytes (one table entry per scheduled row of every sparse column) and the same blob comes back proof
// after proof: its digest is kept per ctx next to a copy of the blob (compared word for word: a cache hit is a memcmp, not a trust decision).
void program_digest(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, uint8_t dg[32]) {
    if (program_words < (1u << 14)) {                     // small statements: hashing is cheaper than remembering
        Sha256::digest((const uint8_t *)h_program, program_words * 8, dg);
        return;
    }
    for (auto &e : ctx->digest_cache)
        if (e.words.size() == program_words && memcmp(e.words.data(), h_program, program_words * 8) == 0) {
            memcpy(dg, e.dg, 32);
            return;
        }
    Sha256::digest((const uint8_t *)h_program, program_words * 8, dg);
    if (ctx->digest_cache.size() >= 4) ctx->digest_cache.erase(ctx->digest_cache.begin());
    ctx->digest_cache.emplace_back();
    ctx->digest_cache.back().words.assign(h_program, h_program + program_words);
    memcpy(ctx->digest_cache.back().dg, dg, 32);
}

// ---- the Fiat-Shamir sponges of stark/transcript.py.  Goldilocks mode: Poseidon-12, rate 8 / capacity 4, on zp_poseidon_sponge.
// BN128 mode: Poseidon-BN254 of width 17 (element 0 = capacity, 1..16 = rate) on zp_poseidon_bn254_sponge; Goldilocks values are
// absorbed three to a field element (every absorb call padded on its own), a Merkle root is one element, and the challenges
// are the three low 64-bit words of the rate elements, each reduced mod p.
struct Transcript {
    zp_ctx *ctx;
    bool bn;
    u64 state[17 * 4];                 // GL: 12 words; BN: 17 elements of 4 words (standard form)
    std::vector<u64> pending, out;     // GL: values; BN: pending holds 4 words per element
    std::vector<u64> log_chal;         // BN: rate element 1 after every flush (4 words each): what the challenge squeezed there is read from (wrap stage B-2)
    std::vector<u64> log_blocks, last_rates, log_caps;   // BN: every absorbed block (16 elements x 4 words) in order; the rate elements of the latest flush; the capacity after every permutation (the wrap circuit's transcript gadgets: zp_wrap_assign)
    int32_t rc = ZP_OK;
    Transcript(zp_ctx *c, bool bn_) : ctx(c), bn(bn_) { memset(state, 0, sizeof state); }
    void absorb(const u64 *v, size_t n) {
        if (!bn) {
            for (size_t i = 0; i < n; i++) pending.push_back(v[i] % GL_P);
        } else {
            for (size_t i = 0; i < n; i += 3) {
                pending.push_back(v[i] % GL_P);
                pending.push_back(i + 1 < n ? v[i + 1] % GL_P : 0);
                pending.push_back(i + 2 < n ? v[i + 2] % GL_P : 0);
                pending.push_back(0);
            }
        }
        out.clear();
    }
    void absorb(const std::vector<u64> &v) { absorb(v.data(), v.size()); }
    void absorb_root(const u64 *root4) {
        if (!bn) { absorb(root4, 4); return; }
        for (int k = 0; k < 4; k++) pending.push_back(root4[k]);
        out.clear();
    }
    void flush(size_t want) {
        const size_t per = bn ? 48 : 8;                    // challenge values one permutation yields
        const size_t extra = want > per ? (want + per - 1) / per - 1 : 0;
        if (!bn) {
            const size_t nblk = (pending.size() + 7) / 8;
            std::vector<u64> blocks(nblk * 8, 0);
            memcpy(blocks.data(), pending.data(), pending.size() * 8);
            pending.clear();
            std::vector<u64> rates((1 + extra) * 8);
            const int32_t r = zp_poseidon_sponge(ctx, (uint64_t *)state, (const uint64_t *)blocks.data(), nblk, extra, (uint64_t *)rates.data());
            if (r != ZP_OK && rc == ZP_OK) rc = r;
            out.assign(rates.begin(), rates.end());
        } else {
            const size_t nel = pending.size() / 4, nblk = (nel + 15) / 16;
            std::vector<u64> blocks(nblk * 64, 0);
            memcpy(blocks.data(), pending.data(), pending.size() * 8);
            pending.clear();
            std::vector<u64> rates((1 + extra) * 64), caps(((nblk ? nblk : 1) + extra) * 4);
            const int32_t r = zp_poseidon_bn254_sponge_caps(ctx, (uint64_t *)state, (const uint64_t *)blocks.data(), nblk, extra, (uint64_t *)rates.data(),
                                                            (uint64_t *)caps.data());
            if (r != ZP_OK && rc == ZP_OK) rc = r;
            log_blocks.insert(log_blocks.end(), blocks.begin(), blocks.end());
            log_caps.insert(log_caps.end(), caps.begin(), caps.end());
            log_chal.insert(log_chal.end(), rates.begin(), rates.begin() + 4);
            last_rates = rates;
            out.clear();
            for (size_t e = 0; e < rates.size() / 4; e++)
                for (int k = 0; k < 3; k++) out.push_back(rates[4 * e + k] % GL_P);
        }
    }
    std::vector<u64> squeeze(size_t n) {
        std::vector<u64> res;
        size_t at = 0;
        while (res.size() < n) {
            if (!pending.empty() || at >= out.size()) {
                flush(n - res.size());
                at = 0;
                if (rc != ZP_OK) { res.resize(n, 0); return res; }
            }
            res.push_back(out[at++]);
        }
        out.erase(out.begin(), out.begin() + at);
        return res;
    }
    e3 challenge() {
        const std::vector<u64> v = squeeze(3);
        return e3_make(v[0], v[1], v[2]);
    }
};

// ---- Merkle trees of the two hash modes behind one set of calls
struct Trees {
    zp_ctx *ctx;
    bool bn;
    static size_t levels16(size_t M) { size_t l = 0; for (size_t n = M; n > 1; n = (n + 15) / 16) l++; return l; }
    size_t tree_words(size_t M) const { return bn ? zp_merkle16_nodes(M) * 4 : (2 * M - 1) * 4; }
    int32_t commit(const u64 *cols, size_t M, int W, u64 *tree) const {
        return bn ? zp_merkle16_commit_bn254(ctx, (const uint64_t *)cols, M, W, (uint64_t *)tree)
                  : zp_merkle_commit(ctx, (const uint64_t *)cols, M, W, (uint64_t *)tree);
    }
    int32_t root(const u64 *tree, size_t M, u64 *out4) const {
        return zp_d2h(ctx, out4, tree + tree_words(M) - 4, 32);           // both layouts end with the root
    }
    size_t path_words(size_t M) const {          // per query
        if (bn) return levels16(M) * 64;
        size_t d = 0;
        while (((size_t)1 << d) < M) d++;
        return (d ? d : 1) * 4;
    }
    int32_t open(const u64 *tree, size_t M, const u64 *idx, int nq, u64 *out) const {
        return bn ? zp_merkle16_open_batch_bn254(ctx, (const uint64_t *)tree, M, (const uint64_t *)idx, nq, (uint64_t *)out)
                  : zp_merkle_open_batch(ctx, (const uint64_t *)tree, M, (const uint64_t *)idx, nq, (uint64_t *)out);
    }
};

// the binary openings record of a BN128-mode proof ("PZOPEN02": zp_stark_openings documents the layout), kept on the ctx.  trees: trace,
// quotient, [stage 2], FRI layers
struct TreeOut { size_t width, rows; const u64 *root; const std::vector<u64> *vals, *paths; size_t pw; };
void openings_record(zp_ctx *ctx, const Transcript &tr, const std::vector<u64> &qidx, int logm, const std::vector<TreeOut> &trees) {
    std::vector<u64> &rec = ctx->last_openings;
    rec.clear();
    rec.insert(rec.end(), {0x33304e45504f5a50ULL /* "PZOPEN03" */, (u64)qidx.size(), (u64)trees.size(), (u64)logm});
    for (const TreeOut &t : trees) rec.insert(rec.end(), {(u64)t.width, (u64)t.rows, (u64)Trees::levels16(t.rows)});
    for (const TreeOut &t : trees) rec.insert(rec.end(), t.root, t.root + 4);
    for (size_t i = 0; i < qidx.size(); i++) {
        rec.push_back(qidx[i]);
        for (const TreeOut &t : trees) {
            rec.insert(rec.end(), t.vals->begin() + i * t.width, t.vals->begin() + (i + 1) * t.width);
            rec.insert(rec.end(), t.paths->begin() + i * t.pw, t.paths->begin() + (i + 1) * t.pw);
        }
    }
    // the transcript (round 5: the wrap circuit hashes it too): every absorbed block, then the rate elements the indices were read from
    rec.push_back((u64)(tr.log_blocks.size() / 64));
    rec.push_back((u64)(tr.last_rates.size() / 64));
    rec.insert(rec.end(), tr.log_blocks.begin(), tr.log_blocks.end());
    rec.insert(rec.end(), tr.last_rates.begin(), tr.last_rates.end());
    rec.insert(rec.end(), tr.log_caps.begin(), tr.log_caps.end());         // one per permutation: n_blocks + (n_rates - 1)
    rec.push_back((u64)(tr.log_chal.size() / 4));                           // "PZOPEN03": the rate element behind every challenge, in squeeze order
    rec.insert(rec.end(), tr.log_chal.begin(), tr.log_chal.end());
}

// Device buffers of one proof.  They come from, and go back to, a per-ctx pool keyed by size: everything runs on the ctx
// stream, so a buffer handed out again is only touched by work enqueued after its previous user.
#define PROVE_POOL_CAP ((size_t)8 << 30)    /* per ctx: a 2^20 x 76 proof keeps ~3 GiB; 8 proving ctxs share one 288 GB GPU */
struct DevBufs {
    zp_ctx *ctx;
    std::vector<std::pair<void *, size_t>> bufs;
    explicit DevBufs(zp_ctx *c) : ctx(c) {}
    ~DevBufs() { for (auto &b : bufs) give_back(b.first, b.second); }
    void give_back(void *p, size_t bytes) {
        if (ctx->prove_pool_bytes + bytes <= PROVE_POOL_CAP) {
            ctx->prove_pool.emplace(bytes, p);
            ctx->prove_pool_bytes += bytes;
        } else {
            (void)zp_dev_free(ctx, p);
        }
    }
    int32_t alloc(size_t elems, u64 **out) {
        const size_t bytes = (elems ? elems : 1) * 8;
        auto it = ctx->prove_pool.find(bytes);
        void *p = nullptr;
        if (it != ctx->prove_pool.end()) {
            p = it->second;
            ctx->prove_pool.erase(it);
            ctx->prove_pool_bytes -= bytes;
        } else {
            int32_t r = zp_dev_alloc(ctx, bytes, &p);
            if (r == ZP_ERR_NOMEM && !ctx->prove_pool.empty()) {      // give the pool back to the device and try once more
                for (auto &kv : ctx->prove_pool) (void)zp_dev_free(ctx, kv.second);
                ctx->prove_pool.clear();
                ctx->prove_pool_bytes = 0;
                r = zp_dev_alloc(ctx, bytes, &p);
            }
            if (r != ZP_OK) return r;
        }
        bufs.emplace_back(p, bytes);
        *out = (u64 *)p;
        return ZP_OK;
    }
    void release(u64 *p) {
        for (size_t i = 0; i < bufs.size(); i++)
            if (bufs[i].first == (void *)p) { give_back(bufs[i].first, bufs[i].second); bufs.erase(bufs.begin() + i); return; }
    }
    void forget(u64 *p) {      // ownership moves elsewhere (a ctx-level cache)
        for (size_t i = 0; i < bufs.size(); i++)
            if (bufs[i].first == (void *)p) { bufs.erase(bufs.begin() + i); return; }
    }
};

// ---- JSON text exactly as json.dumps(proof, separators=(",", ":")) writes it
// a chunk proof's text is ~150 000 decimal numbers: two digits per division from a table instead of snprintf (6.4 -> ~2 ms of a 90 ms proof at 2^22 rows)
void j_u64(std::string &s, u64 v) {
    static const char D2[201] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
        "707172737475767778798081828384858687888990919293949596979899";
    char b[24];
    int at = 24;
    while (v >= 100) {
        const unsigned r = (unsigned)(v % 100);
        v /= 100;
        b[--at] = D2[2 * r + 1];
        b[--at] = D2[2 * r];
    }
    if (v >= 10) {
        b[--at] = D2[2 * v + 1];
        b[--at] = D2[2 * v];
    } else {
        b[--at] = (char)('0' + v);
    }
    s.append(b + at, (size_t)(24 - at));
}
void j_list(std::string &s, const u64 *v, size_t n) {
    s += '[';
    for (size_t i = 0; i < n; i++) { if (i) s += ','; j_u64(s, v[i]); }
    s += ']';
}
void j_e3list(std::string &s, const std::vector<u64> &v) {   // [[a,b,c],...]
    s += '[';
    for (size_t i = 0; i * 3 < v.size(); i++) { if (i) s += ','; j_list(s, &v[3 * i], 3); }
    s += ']';
}
void j_dec256(std::string &s, const u64 *w4) {     // "decimal" of a 256-bit little-endian value, quoted
    // 19 digits at a time (10^19 < 2^64): five long divisions instead of 78 -- a BN128-mode proof text carries ~20 000 such numbers
    // (16 digests per level of every authentication path), and digit-by-digit they were 8 ms of a 93 ms final STARK
    constexpr u64 TEN19 = 10000000000000000000ULL;
    u64 v[4] = {w4[0], w4[1], w4[2], w4[3]};
    u64 chunk[5];
    int n = 0;
    while (v[0] | v[1] | v[2] | v[3]) {
        unsigned __int128 rem = 0;
        for (int k = 3; k >= 0; k--) {
            const unsigned __int128 cur = (rem << 64) | v[k];
            v[k] = (u64)(cur / TEN19);
            rem = cur % TEN19;
        }
        chunk[n++] = (u64)rem;
    }
    s += '"';
    if (!n) {
        s += '0';
    } else {
        char b[24];
        snprintf(b, sizeof b, "%llu", (unsigned long long)chunk[n - 1]);
        s += b;
        for (int k = n - 2; k >= 0; k--) {
            snprintf(b, sizeof b, "%019llu", (unsigned long long)chunk[k]);
            s += b;
        }
    }
    s += '"';
}
void j_root(std::string &s, const u64 *root4, bool bn) {
    if (!bn) { j_list(s, root4, 4); return; }
    s += '[';
    j_dec256(s, root4);
    s += ']';
}
void j_opening_bn(std::string &s, const u64 *vals, size_t W, const u64 *path, size_t levels) {   // path: per level 16 digests as strings
    s += "{\"values\":";
    j_list(s, vals, W);
    s += ",\"path\":[";
    for (size_t l = 0; l < levels; l++) {
        if (l) s += ',';
        s += '[';
        for (int c = 0; c < 16; c++) { if (c) s += ','; j_dec256(s, path + (l * 16 + c) * 4); }
        s += ']';
    }
    s += "]}";
}
void j_opening(std::string &s, const u64 *vals, size_t W, const u64 *path, size_t depth) {   // {"values":[..],"path":[[4]..]}
    s += "{\"values\":";
    j_list(s, vals, W);
    s += ",\"path\":[";
    for (size_t d = 0; d < depth; d++) { if (d) s += ','; j_list(s, path + 4 * d, 4); }
    s += "]}";
}

#define PV_TRY(expr)                                                    \
    do {                                                                \
        const int32_t rc_ = (expr);                                     \
        if (rc_ != ZP_OK) return rc_;                                   \
    } while (0)

}  // namespace

int32_t zpi_pool_alloc(zp_ctx *ctx, size_t bytes, void **out) {
    DevBufs b(ctx);
    u64 *p = nullptr;
    const int32_t rc = b.alloc((bytes + 7) / 8, &p);
    if (rc != ZP_OK) return rc;
    b.forget(p);
    *out = p;
    return ZP_OK;
}
void zpi_pool_release(zp_ctx *ctx, void *p, size_t bytes) {
    DevBufs b(ctx);
    b.give_back(p, ((bytes + 7) / 8 ? (bytes + 7) / 8 : 1) * 8);
}
void zpi_sha256(const uint8_t *data, size_t len, uint8_t *out32) { Sha256::digest(data, len, out32); }

extern "C" {

int32_t zp_free_buffer(void *p) {
    free(p);
    return ZP_OK;
}

// A generated constraint kernel (AIR plug-in ABI: stark/air.py writes it, `zpair_<air>_quotient` in its own shared library) for the one-call
// provers of THIS ctx: proofs of the program with this digest evaluate their constraints through it instead of the interpreter -- the same
// values (whole proofs are byte-identical whichever evaluator ran), 0.7 instead of 1.2 ms at 2^21 x 76.  fn = NULL forgets it.  Programs with
// sparse periodic fixed columns stay with the interpreter (the generated kernels do not read them), and so do sharded proofs (row windows).
typedef int (*zp_air_quotient_fn)(void *stream, const u64 *cols, const u64 *fixedc, u64 M, u64 b, const u64 *pub, const u64 *apow, const u64 *zhinv,
                                  const u64 *xs_lo, const u64 *xs_hi, int lb, u64 shift, u64 wlast, u64 *out);
static std::string digest_hex64(const uint64_t *h_program, size_t program_words) {
    uint8_t dg[32];
    Sha256::digest((const uint8_t *)h_program, program_words * 8, dg);
    char hex[65];
    for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", dg[i]);
    return std::string(hex, 64);
}
int32_t zp_stark_set_air_kernel(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, void *quotient_fn) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, h_program && program_words >= 8 && program_words < ((size_t)1 << 28), "null / implausible program");
    try {
        const std::string k = digest_hex64(h_program, program_words);
        if (quotient_fn) ctx->air_kernels[k] = quotient_fn;
        else ctx->air_kernels.erase(k);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
    return ZP_OK;
}
// The ROW-WINDOW form of a generated constraint kernel (`zpair_<air>_quotient_rows` of the same library; round 6): the sharded provers of this ctx
// (zp_stark_prove_sharded, zp_stark_prove_sharded_bn128: every rank evaluates the quotient on ITS rows) use it instead of the interpreter -- same
// values, proofs byte-identical either way.  fn = NULL forgets it.
typedef int (*zp_air_quotient_rows_fn)(void *stream, const u64 *cols, u64 sc, const u64 *fixedc, u64 sf, u64 M, u64 b, u64 row0, u64 nrows, const u64 *pub,
                                       const u64 *apow, const u64 *zhinv, const u64 *xs_lo, const u64 *xs_hi, int lb, u64 shift, u64 wlast, u64 *out, u64 so);
int32_t zp_stark_set_air_kernel_rows(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, void *quotient_rows_fn) {
    if (!ctx) return ZP_ERR_ARG;
    ZP_ARG(ctx, h_program && program_words >= 8 && program_words < ((size_t)1 << 28), "null / implausible program");
    try {
        const std::string k = "rows:" + digest_hex64(h_program, program_words);
        if (quotient_rows_fn) ctx->air_kernels[k] = quotient_rows_fn;
        else ctx->air_kernels.erase(k);
    } catch (...) {
        ctx->err = "out of host memory";
        return ZP_ERR_NOMEM;
    }
    return ZP_OK;
}

// SHA-256 of a constraint program blob: the AIR digest.  out32 = the 32 digest bytes (a proof text names the first 8 as 16 hex digits);
// out_words4 (may be NULL) = the four little-endian 64-bit words, each reduced mod p, that the provers absorb into the transcript.
int32_t zp_program_digest(const uint64_t *h_program, size_t program_words, uint8_t *out32, uint64_t *out_words4) {
    if (!h_program || !out32 || program_words == 0) return ZP_ERR_ARG;
    Sha256::digest((const uint8_t *)h_program, program_words * 8, out32);
    if (out_words4)
        for (int i = 0; i < 4; i++) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; b--) w = (w << 8) | out32[8 * i + b];
            out_words4[i] = w % GL_P;
        }
    return ZP_OK;
}

// The fixed columns of a program on its evaluation domain (zp_fixed_columns: boundary selectors + one extended period of every sparse periodic
// column), from the ctx's cache: they depend on the domain and the program only; columns that hold public inputs (expected roots / indices /
// transcript words of a verifier AIR: 37 of 104 at the service's size) are refreshed in place per proof, the others stay.  The buffer belongs
// to the ctx.  (Both provers: the sharded one cuts its row windows out of it.)
static int32_t fixed_columns_cached(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const std::vector<ZpFixedCol> &fxc, const uint64_t *h_pubs,
                             int32_t n_pubs, int32_t logn, int32_t logb, u64 shift, u64 root32, const char *dg_hex, u64 **out) {
    bool has_pub = false;
    for (const ZpFixedCol &fc : fxc) has_pub |= fc.has_pub;
    const size_t fwords = zp_fixed_columns_words(h_program, program_words, logn, logb);
    ZP_ARG(ctx, fwords != 0, "fixed column longer than the trace");
    char key[128];
    snprintf(key, sizeof key, "%d/%d/%llx/%llx/%s", logn, logb, (unsigned long long)shift, (unsigned long long)root32, fxc.empty() ? "" : dg_hex);
    auto it = ctx->prove_fixed.find(key);
    if (it != ctx->prove_fixed.end()) {
        *out = it->second;
        if (has_pub) {
            ZpStage stage_fx(ctx, "fixed_columns");
            PV_TRY(zpi_fixed_columns_build(ctx, h_program, program_words, h_pubs, n_pubs, logn, logb, shift, (uint64_t *)*out, fwords, true));
        }
        return ZP_OK;
    }
    // (a verifier AIR's columns are gigabytes -- 91 full-length columns at the service's size: 3 GB --; a ctx that has met many shapes
    // starts over rather than grow without bound)
    if (ctx->prove_fixed_bytes + fwords * 8 > ((size_t)24 << 30) && !ctx->prove_fixed.empty()) {
        PV_TRY(zp_sync(ctx));
        for (auto &kv : ctx->prove_fixed) (void)hipFree(kv.second);
        ctx->prove_fixed.clear();
        ctx->prove_fixed_bytes = 0;
    }
    void *pf = nullptr;
    PV_TRY(zp_dev_alloc(ctx, fwords * 8, &pf));
    const int32_t r = zp_fixed_columns(ctx, h_program, program_words, h_pubs, n_pubs, logn, logb, shift, (uint64_t *)pf, fwords);
    if (r != ZP_OK) { (void)zp_dev_free(ctx, pf); return r; }
    ctx->prove_fixed[key] = (u64 *)pf;
    ctx->prove_fixed_bytes += fwords * 8;
    *out = (u64 *)pf;
    return ZP_OK;
}

static int32_t prove_impl(zp_ctx *ctx, bool bn, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                          size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf, int32_t fri_final_log,
                          int32_t n_queries, int32_t pow_bits, char **out_json, size_t *out_len) {
    ZpStage stage_(ctx, bn ? "stark_prove_bn128" : "stark_prove");
    ZP_ARG(ctx, air_name && h_program && d_trace && out_json && out_len && (h_pubs || n_pubs == 0), "null pointer");
    {   // the name goes into the proof text verbatim: letters, digits, '_', '-', '.' only
        const size_t nl = strlen(air_name);
        bool ok = nl >= 1 && nl <= 64;
        for (size_t i = 0; ok && i < nl; i++) {
            const char ch = air_name[i];
            ok = (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_' || ch == '-' || ch == '.';
        }
        ZP_ARG(ctx, ok, "air_name must be 1..64 characters of [A-Za-z0-9_.-]");
    }
    ZP_ARG(ctx, program_words >= 12, "constraint program shorter than its header");
    static const unsigned char magic[8] = {'Z', 'P', 'A', 'I', 'R', '1', 0, 0};
    ZP_ARG(ctx, memcmp(h_program, magic, 8) == 0, "not a ZPAIR1 constraint program");
    const size_t W = h_program[1], W2 = h_program[2], n_pub_prog = h_program[4], n_chal = h_program[5], n_const = h_program[6],
                 n_instr = h_program[7], K = h_program[8], n_s2 = h_program[10], Q = h_program[11];
    std::vector<ZpFixedCol> fxc;
    ZP_ARG(ctx, n_const < (1u << 16) && n_instr < (1u << 24) && n_s2 < (1u << 16) && zpi_program_fixed_table(h_program, program_words, &fxc),
           "constraint program length does not match its header");
    ZP_ARG(ctx, (size_t)n_pubs == n_pub_prog, "number of public inputs does not match the program");
    ZP_ARG(ctx, W >= 1 && W < 4096 && W2 < 4096 && K >= 1 && Q >= 1 && Q <= 16, "program dimensions out of range");
    ZP_ARG(ctx, logn >= 1 && logb >= 1 && logn + logb <= 30 && fri_logf >= 1 && fri_logf <= 4 && fri_final_log >= 0 && fri_final_log < logn &&
                    n_queries >= 1 && n_queries <= 4096 && pow_bits >= 0 && pow_bits <= 40, "STARK parameters out of range");
    ZP_ARG(ctx, Q <= ((size_t)1 << logb), "the blow-up must cover the quotient degree");
    ZP_ARG(ctx, trace_words == (W << logn), "trace_words must be W * 2^logn (W from the program header)");
    ZP_ARG(ctx, (n_s2 == 0) == (W2 == 0) && (n_s2 == 0 || n_chal == 3), "stage-2 table and widths disagree");
    for (int i = 0; i < n_pubs; i++) ZP_ARG(ctx, h_pubs[i] < GL_P, "public input not canonical");
    const u64 *stage2 = (const u64 *)h_program + 12 + n_const + n_instr;
    size_t w2sum = 0;
    for (size_t k = 0; k < n_s2; k++) {
        const u64 kind = stage2[4 * k];
        ZP_ARG(ctx, kind == 1 || kind == 2, "unknown stage-2 argument");
        ZP_ARG(ctx, stage2[4 * k + 1] < W && stage2[4 * k + 2] < W && stage2[4 * k + 3] < W, "stage-2 column out of range");
        w2sum += kind == 1 ? 3 : 9;
    }
    ZP_ARG(ctx, w2sum == W2, "stage-2 width does not match its table");

    const int logm = logn + logb;
    const size_t N = (size_t)1 << logn, M = (size_t)1 << logm, Wt = W + W2;
    const u64 shift = ctx->coset_shift, root32 = ctx->root32;
    const u64 wN = gl_root(root32, logn);
    DevBufs dev(ctx);
    // measurement aid (ZP_PROVE_TRACE=1): host-clock milliseconds since entry at the stage boundaries, the stream drained at each -- where the
    // wall time of a proof goes that the per-entry-point GPU times do not show
    static const bool trace_on = getenv("ZP_PROVE_TRACE") != nullptr;
    const auto t_entry = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(ctx->stream);
        fprintf(stderr, "[prove %s 2^%d] %-28s %8.3f ms\n", air_name, logn, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_entry).count());
    };

    // AIR digest: sha256 of the blob; the first 16 hex digits name it, four little-endian words go into the transcript
    uint8_t dg[32];
    program_digest(ctx, h_program, program_words, dg);
    char dg_hex[17];
    for (int i = 0; i < 8; i++) snprintf(dg_hex + 2 * i, 3, "%02x", dg[i]);
    std::vector<u64> first = {(u64)logn, (u64)logb, (u64)W, (u64)W2, (u64)fri_logf, (u64)fri_final_log, (u64)n_queries, (u64)pow_bits, root32, shift};
    for (int i = 0; i < 4; i++) {
        u64 wd = 0;
        for (int k = 0; k < 8; k++) wd |= (u64)dg[8 * i + k] << (8 * k);
        first.push_back(wd % GL_P);
    }
    first.push_back((u64)n_pubs);
    Transcript tr(ctx, bn);
    const Trees T{ctx, bn};
    if (n_pubs <= 64) {
        for (int i = 0; i < n_pubs; i++) first.push_back(h_pubs[i]);
        tr.absorb(first);
    } else {
        // a long public-input vector (a verifier AIR: every root, index and opened value of its inner proofs) enters the transcript
        // as ONE commitment instead of thousands of dependent sponge permutations: Goldilocks mode: rows of 8 values (zero padded,
        // row count a power of two >= 2), binary Poseidon tree;  BN128 mode: rows of 48 values, column-major, 16-ary tree
        tr.absorb(first);
        size_t Mp;
        std::vector<u64> mat;
        if (!bn) {
            Mp = 2;
            while (Mp * 8 < (size_t)n_pubs) Mp <<= 1;
            mat.assign(Mp * 8, 0);
            for (int i = 0; i < n_pubs; i++) mat[i] = h_pubs[i];
        } else {
            Mp = ((size_t)n_pubs + 47) / 48;
            mat.assign(Mp * 48, 0);
            for (int i = 0; i < n_pubs; i++) mat[(size_t)(i % 48) * Mp + (size_t)(i / 48)] = h_pubs[i];
        }
        u64 *dmat, *dtree;
        PV_TRY(dev.alloc(mat.size(), &dmat));
        PV_TRY(dev.alloc(T.tree_words(Mp), &dtree));
        PV_TRY(zp_h2d(ctx, dmat, mat.data(), mat.size() * 8));
        if (!bn) PV_TRY(zp_merkle_commit_rows(ctx, (const uint64_t *)dmat, Mp, 8, (uint64_t *)dtree));
        else PV_TRY(T.commit(dmat, Mp, 48, dtree));
        u64 rootp[4];
        PV_TRY(T.root(dtree, Mp, rootp));
        dev.release(dmat);
        dev.release(dtree);
        tr.absorb_root(rootp);
    }

    mark("transcript head");
    // 1. commit the trace (ext has room for the stage-2 columns behind the trace columns).  No coefficient buffer since round 5: the
    //    out-of-domain evaluations come from the resident extension (zp_ood_eval), so the extensions run without their coefficient
    //    store -- on the fused seam kernel where the plan allows it (csrc/ntt.hip) -- and W N 8 bytes per proof are never written
    u64 *ext, *tree1;
    PV_TRY(dev.alloc(Wt * M, &ext));
    // BN128 mode: a leaf holds 2^g rows i, i + M', ... of its tree (the column-major matrix reinterpreted as [width 2^g][M'], like a
    // FRI layer), g the largest with width 2^g <= 56 values = one width-17 permutation per leaf (stark/prover.py:
    // bn128_rows_per_leaf_log; Goldilocks mode: g = 0)
    auto rows_per_leaf_log = [&](size_t width) {
        int g = 0;
        if (bn && width)
            while ((width << (g + 1)) <= 56 && g + 1 <= logm - 4) g++;
        return g;
    };
    const int gt = rows_per_leaf_log(W), g2 = rows_per_leaf_log(W2);
    const size_t Mt = M >> gt, Wtg = W << gt, M2 = M >> g2, W2g = W2 << g2;
    PV_TRY(dev.alloc(T.tree_words(Mt), &tree1));
    PV_TRY(zp_lde(ctx, d_trace, (uint64_t *)ext, nullptr, logn, logb, (int32_t)W, shift));
    PV_TRY(T.commit(ext, Mt, (int)Wtg, tree1));
    u64 root1[4], root2[4] = {0, 0, 0, 0}, rootq[4];
    PV_TRY(T.root(tree1, Mt, root1));
    tr.absorb_root(root1);
    std::vector<u64> pubchal(h_pubs, h_pubs + n_pubs);
    u64 *tree2 = nullptr, *s2_kept = nullptr;
    if (n_s2) {
        const e3 chal = tr.challenge();
        PV_TRY(tr.rc);
        u64 *s2;
        PV_TRY(dev.alloc(W2 * N, &s2));
        size_t at = 0;
        for (size_t k = 0; k < n_s2; k++) {
            const u64 *st = stage2 + 4 * k;
            if (st[0] == 1) {
                PV_TRY(zp_grand_product(ctx, d_trace + st[1] * N, d_trace + st[2] * N, N, (const uint64_t *)chal.c, (uint64_t *)(s2 + at * N)));
                at += 3;
            } else {
                PV_TRY(zp_logup_columns(ctx, d_trace + st[1] * N, d_trace + st[2] * N, d_trace + st[3] * N, N, (const uint64_t *)chal.c, (uint64_t *)(s2 + at * N)));
                at += 9;
            }
        }
        PV_TRY(zp_lde(ctx, (const uint64_t *)s2, (uint64_t *)(ext + W * M), nullptr, logn, logb, (int32_t)W2, shift));
        PV_TRY(dev.alloc(T.tree_words(M2), &tree2));
        PV_TRY(T.commit(ext + W * M, M2, (int)W2g, tree2));
        PV_TRY(T.root(tree2, M2, root2));
        s2_kept = s2;                         // the stage-2 columns on the trace domain: read once more for their out-of-domain evaluations
        tr.absorb_root(root2);
        for (int c = 0; c < 3; c++) pubchal.push_back(chal.c[c]);
    }
    const e3 alpha = tr.challenge();
    PV_TRY(tr.rc);

    mark("trace (+stage 2) committed");
    // 2. constraint quotient on the coset
    u64 *fixed, *dq, *dqcoef;
    PV_TRY(fixed_columns_cached(ctx, h_program, program_words, fxc, h_pubs, n_pubs, logn, logb, shift, root32, dg_hex, &fixed));
    mark("fixed columns");
    std::vector<u64> apow(3 * K);
    {
        e3 cur = e3_make(1, 0, 0);
        for (size_t k = 0; k < K; k++) { memcpy(&apow[3 * k], cur.c, 24); cur = e3_mul(cur, alpha); }
    }
    std::vector<u64> zhinv((size_t)1 << logb);
    {
        const u64 sN = gl_pow(shift, (u64)N), wb = gl_root(root32, logb);
        u64 p = 1;
        for (size_t j = 0; j < zhinv.size(); j++) { zhinv[j] = gl_inv(gl_sub(gl_mul(sN, p), 1)); p = gl_mul(p, wb); }
    }
    PV_TRY(dev.alloc(3 * M, &dq));
    zp_air_quotient_fn plug = nullptr;
    if (!ctx->air_kernels.empty()) {       // (round 5: generated kernels read the sparse periodic fixed columns too -- one extended period each, zp_fixed_columns' layout)
        // (the digest computed above -- through the per-ctx cache, a memcmp for a program seen before -- not a second SHA-256 of the blob:
        // for a verifier AIR's 7 MB that second hash was 16 ms of every recursion STARK, round 5)
        char hex[65];
        for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", dg[i]);
        auto it = ctx->air_kernels.find(std::string(hex, 64));
        if (it != ctx->air_kernels.end()) plug = (zp_air_quotient_fn)it->second;
    }
    if (plug) {
        // the generated kernel of this program (zp_stark_set_air_kernel): its small operands go up in one buffer [pub | 0 | apow | zhinv]
        std::vector<u64> ops(pubchal);
        ops.push_back(0);
        const size_t o_ap = ops.size();
        ops.insert(ops.end(), apow.begin(), apow.end());
        const size_t o_zh = ops.size();
        ops.insert(ops.end(), zhinv.begin(), zhinv.end());
        u64 *d_ops;
        PV_TRY(dev.alloc(ops.size(), &d_ops));
        PV_TRY(zp_h2d(ctx, d_ops, ops.data(), ops.size() * 8));
        const uint64_t *xlo, *xhi;
        int32_t xlb;
        PV_TRY(zp_domain_tables(ctx, logm, &xlo, &xhi, &xlb));
        const int hrc = plug((void *)ctx->stream, (const u64 *)ext, (const u64 *)fixed, (u64)M, (u64)1 << logb, d_ops, d_ops + o_ap, d_ops + o_zh, (const u64 *)xlo,
                             (const u64 *)xhi, (int)xlb, shift, gl_inv(wN), dq);
        if (hrc != 0) {
            ctx->err = "generated constraint kernel: launch failed (hip error " + std::to_string(hrc) + ")";
            return ZP_ERR_HIP;
        }
        PV_TRY(zp_sync(ctx));
        dev.release(d_ops);
    } else {
        PV_TRY(zp_eval_quotient(ctx, h_program, program_words, (const uint64_t *)ext, (const uint64_t *)fixed, logm, logb, (const uint64_t *)pubchal.data(),
                                (int32_t)pubchal.size(), (const uint64_t *)apow.data(), (const uint64_t *)zhinv.data(), shift, gl_inv(wN), (uint64_t *)dq));
    }
    mark("quotient evaluated");
    int q_logn = logm;
    size_t Wq = 3;
    u64 *treeq;
    PV_TRY(dev.alloc(T.tree_words(M), &treeq));
    if (Q > 1) {
        // q(x) = sum_j (x / shift)^(jN) qt_j(x): the pieces are slices of the coefficient vector of q(shift X) (c_i shift^i); their LDEs get
        // committed.  (Q == 1: the quotient is committed as it stands and never leaves the evaluation form.)
        u64 *pad, *pext;
        PV_TRY(dev.alloc(3 * M, &dqcoef));
        PV_TRY(zp_intt(ctx, (const uint64_t *)dq, (uint64_t *)dqcoef, logm, 3));
        PV_TRY(dev.alloc(3 * Q * M, &pad));
        PV_TRY(zp_dev_zero(ctx, pad, 3 * Q * M * 8));
        for (size_t j = 0; j < Q; j++)
            for (int c = 0; c < 3; c++) PV_TRY(zp_d2d(ctx, pad + (3 * j + c) * M, dqcoef + c * M + j * N, N * 8));
        PV_TRY(dev.alloc(3 * Q * M, &pext));
        PV_TRY(zp_ntt(ctx, (const uint64_t *)pad, (uint64_t *)pext, logm, (int32_t)(3 * Q)));
        PV_TRY(zp_sync(ctx));
        dev.release(pad);
        dev.release(dq);
        dev.release(dqcoef);
        dq = pext;
        q_logn = logn;
        Wq = 3 * Q;
    }
    // BN128 mode: 2^qg rows of the quotient per leaf (rows i, i + M', ...: the matrix reinterpreted as [Wq 2^qg][M'], like a FRI layer),
    // qg the largest with Wq 2^qg <= 48 values = one width-17 permutation per leaf (stark/prover.py: bn128_rows_per_leaf_log)
    const int qg = rows_per_leaf_log(Wq);
    const size_t Mq = M >> qg, Wqg = Wq << qg;
    PV_TRY(T.commit(dq, Mq, (int)Wqg, treeq));
    PV_TRY(T.root(treeq, Mq, rootq));
    tr.absorb_root(rootq);
    const e3 zeta = tr.challenge();
    PV_TRY(tr.rc);

    mark("quotient committed");
    // 3. out-of-domain evaluations FROM VALUES (barycentric form, zp_ood_eval): a polynomial of degree < 2^d is read on a 2^d-point domain it is
    //    known on; the trace and stage-2 columns at zeta and zeta w in ONE pass
    const e3 zeta_w = e3_scale(zeta, wN);
    std::vector<u64> ev_all((Wt + Wq) * 3), ev_next(Wt * 3);
    // (the witness columns are read where they lie CONTIGUOUSLY -- the trace itself and the stage-2 columns, on the trace domain: shift 1, stride 1,
    // 8 N bytes per column; the same numbers as from the 2^logn-point sub-coset of the extension (row stride 2^logb), which costs 16 N)
    PV_TRY(zp_ood_eval(ctx, d_trace, N, 1, (int32_t)W, logn, 1, (const uint64_t *)zeta.c, 1, (uint64_t *)ev_all.data(), (uint64_t *)ev_next.data()));
    if (W2) {
        PV_TRY(zp_ood_eval(ctx, (const uint64_t *)s2_kept, N, 1, (int32_t)W2, logn, 1, (const uint64_t *)zeta.c, 1, (uint64_t *)(ev_all.data() + W * 3),
                           (uint64_t *)(ev_next.data() + W * 3)));
        PV_TRY(zp_sync(ctx));
        dev.release(s2_kept);
    }
    PV_TRY(zp_ood_eval(ctx, (const uint64_t *)dq, M, (size_t)1 << (logm - q_logn), (int32_t)Wq, q_logn, shift, (const uint64_t *)zeta.c, 0,
                       (uint64_t *)(ev_all.data() + Wt * 3), nullptr));
    tr.absorb(ev_all);
    tr.absorb(ev_next);
    const e3 gamma = tr.challenge();
    PV_TRY(tr.rc);

    mark("out-of-domain evaluations");
    // 4. DEEP quotient
    u64 *df;
    PV_TRY(dev.alloc(3 * M, &df));
    PV_TRY(zp_deep_quotient(ctx, (const uint64_t *)ext, (int32_t)Wt, (const uint64_t *)dq, (int32_t)Wq, logm, (int32_t)Wt, (const uint64_t *)zeta.c, (const uint64_t *)zeta_w.c, (const uint64_t *)gamma.c,
                            (const uint64_t *)ev_all.data(), (const uint64_t *)ev_next.data(), shift, (uint64_t *)df));

    // 5. FRI
    struct Layer { int lg, f; u64 *tree, *data; u64 root[4]; };
    std::vector<Layer> layers;
    int cur = logm;
    u64 cur_shift = shift;
    u64 *dlayer = df;
    while (cur > fri_final_log + logb) {
        const int f = fri_logf < cur - (fri_final_log + logb) ? fri_logf : cur - (fri_final_log + logb);
        Layer L;
        L.lg = cur; L.f = f; L.data = dlayer;
        const size_t m = (size_t)1 << (cur - f);
        PV_TRY(dev.alloc(T.tree_words(m), &L.tree));
        PV_TRY(T.commit(dlayer, m, 3 << f, L.tree));   // leaf = the 2^f * 3 values folded together
        PV_TRY(T.root(L.tree, m, L.root));
        tr.absorb_root(L.root);
        const e3 beta = tr.challenge();
        PV_TRY(tr.rc);
        u64 *next;
        PV_TRY(dev.alloc((size_t)3 << (cur - f), &next));
        PV_TRY(zp_fri_fold(ctx, (const uint64_t *)dlayer, (uint64_t *)next, cur, f, (const uint64_t *)beta.c, cur_shift));
        layers.push_back(L);
        dlayer = next;
        cur_shift = gl_pow(cur_shift, (u64)1 << f);
        cur -= f;
    }
    const int final_log = cur;
    std::vector<u64> final_l((size_t)3 << final_log);
    PV_TRY(zp_d2h(ctx, final_l.data(), dlayer, final_l.size() * 8));
    for (int c = 0; c < 3; c++) tr.absorb(&final_l[(size_t)c << final_log], (size_t)1 << final_log);   // plane by plane (BN128 mode pads every call)

    mark("DEEP + FRI");
    // 6. proof of work, then the queries
    u64 nonce = 0;
    if (pow_bits) {
        const std::vector<u64> seed = tr.squeeze(4);
        PV_TRY(tr.rc);
        PV_TRY(zp_pow_grind(ctx, (const uint64_t *)seed.data(), pow_bits, (uint64_t *)&nonce));
        tr.absorb(&nonce, 1);
    }
    std::vector<u64> qidx = tr.squeeze((size_t)n_queries);
    PV_TRY(tr.rc);
    for (u64 &v : qidx) v &= (M - 1);
    const size_t nq = (size_t)n_queries, depth = (size_t)logm;
    const size_t pwq = T.path_words(Mq), pwt = T.path_words(Mt), pw2 = T.path_words(M2);
    std::vector<u64> v_tr(nq * Wtg), p_tr(nq * pwt), v_s2, p_s2, v_q(nq * Wqg), p_q(nq * pwq);
    {
        std::vector<u64> rows = qidx;
        for (u64 &v : rows) v &= (Mt - 1);
        PV_TRY(zp_gather_rows(ctx, (const uint64_t *)ext, Mt, (int32_t)Wtg, (const uint64_t *)rows.data(), n_queries, (uint64_t *)v_tr.data()));
        PV_TRY(T.open(tree1, Mt, rows.data(), n_queries, p_tr.data()));
    }
    if (n_s2) {
        v_s2.resize(nq * W2g);
        p_s2.resize(nq * pw2);
        std::vector<u64> rows = qidx;
        for (u64 &v : rows) v &= (M2 - 1);
        PV_TRY(zp_gather_rows(ctx, (const uint64_t *)(ext + W * M), M2, (int32_t)W2g, (const uint64_t *)rows.data(), n_queries, (uint64_t *)v_s2.data()));
        PV_TRY(T.open(tree2, M2, rows.data(), n_queries, p_s2.data()));
    }
    {
        std::vector<u64> qrows = qidx;
        for (u64 &v : qrows) v &= (Mq - 1);
        PV_TRY(zp_gather_rows(ctx, (const uint64_t *)dq, Mq, (int32_t)Wqg, (const uint64_t *)qrows.data(), n_queries, (uint64_t *)v_q.data()));
        PV_TRY(T.open(treeq, Mq, qrows.data(), n_queries, p_q.data()));
    }
    struct FriOpen { std::vector<u64> vals, paths; size_t width, depth, pw, m; };
    std::vector<FriOpen> fo(layers.size());
    {
        std::vector<u64> pos = qidx;
        for (size_t li = 0; li < layers.size(); li++) {
            const Layer &L = layers[li];
            const size_t m = (size_t)1 << (L.lg - L.f);
            for (u64 &p : pos) p &= (m - 1);
            fo[li].width = (size_t)3 << L.f;
            fo[li].depth = (size_t)(L.lg - L.f);
            fo[li].m = m;
            fo[li].pw = T.path_words(m);
            fo[li].vals.resize(nq * fo[li].width);
            fo[li].paths.resize(nq * fo[li].pw);
            PV_TRY(zp_gather_rows(ctx, (const uint64_t *)L.data, m, (int32_t)fo[li].width, (const uint64_t *)pos.data(), n_queries, (uint64_t *)fo[li].vals.data()));
            PV_TRY(T.open(L.tree, m, pos.data(), n_queries, fo[li].paths.data()));
        }
    }

    if (bn) {       // the same openings in binary, for the Groth16 wrap's witness (zp_stark_openings -> zp_wrap_assign): no text round trip
        std::vector<TreeOut> trees = {{Wtg, Mt, root1, &v_tr, &p_tr, pwt}, {Wqg, Mq, rootq, &v_q, &p_q, pwq}};
        if (n_s2) trees.push_back({W2g, M2, root2, &v_s2, &p_s2, pw2});
        for (size_t li = 0; li < layers.size(); li++) trees.push_back({fo[li].width, fo[li].m, layers[li].root, &fo[li].vals, &fo[li].paths, fo[li].pw});
        openings_record(ctx, tr, qidx, logm, trees);
    }

    mark("queries opened");
    // the proof text
    std::string s;
    s.reserve(nq * (Wt + Wq + 64) * 24 + (1 << 16));
    s += "{\"air\":\"";
    s += air_name;
    s += "\",\"air_digest\":\"";
    s += dg_hex;
    s += "\",\"params\":{\"logn\":";
    j_u64(s, (u64)logn); s += ",\"logb\":"; j_u64(s, (u64)logb); s += ",\"fri_logf\":"; j_u64(s, (u64)fri_logf);
    s += ",\"fri_final_log\":"; j_u64(s, (u64)fri_final_log); s += ",\"n_queries\":"; j_u64(s, (u64)n_queries);
    s += ",\"pow_bits\":"; j_u64(s, (u64)pow_bits);
    if (bn) s += ",\"hash\":\"bn128\"";
    s += "},\"root32\":"; j_u64(s, root32);
    s += ",\"shift\":"; j_u64(s, shift);
    s += ",\"publics\":"; j_list(s, (const u64 *)h_pubs, (size_t)n_pubs);
    s += ",\"roots\":{\"trace\":"; j_root(s, root1, bn);
    s += ",\"quotient\":"; j_root(s, rootq, bn);
    if (n_s2) { s += ",\"stage2\":"; j_root(s, root2, bn); }
    s += "},\"evals\":{\"z\":"; j_e3list(s, ev_all);
    s += ",\"zw\":"; j_e3list(s, ev_next);
    s += "},\"fri\":{\"roots\":[";
    for (size_t li = 0; li < layers.size(); li++) { if (li) s += ','; j_root(s, layers[li].root, bn); }
    s += "],\"final\":[";
    for (int c = 0; c < 3; c++) { if (c) s += ','; j_list(s, &final_l[(size_t)c << final_log], (size_t)1 << final_log); }
    s += "]},\"queries\":[";
    for (size_t i = 0; i < nq; i++) {
        if (i) s += ',';
        s += "{\"index\":"; j_u64(s, qidx[i]);
        auto opening = [&](const u64 *vals, size_t width, const u64 *path, size_t rows, size_t bin_depth) {
            if (bn) j_opening_bn(s, vals, width, path, Trees::levels16(rows));
            else j_opening(s, vals, width, path, bin_depth);
        };
        s += ",\"trace\":"; opening(&v_tr[i * Wtg], Wtg, &p_tr[i * pwt], Mt, depth);
        s += ",\"quotient\":"; opening(&v_q[i * Wqg], Wqg, &p_q[i * pwq], Mq, depth);
        if (n_s2) { s += ",\"stage2\":"; opening(&v_s2[i * W2g], W2g, &p_s2[i * pw2], M2, depth); }
        s += ",\"fri\":[";
        for (size_t li = 0; li < layers.size(); li++) {
            if (li) s += ',';
            opening(&fo[li].vals[i * fo[li].width], fo[li].width, &fo[li].paths[i * fo[li].pw], fo[li].m, fo[li].depth);
        }
        s += "]}";
    }
    s += ']';
    if (pow_bits) { s += ",\"pow_nonce\":"; j_u64(s, nonce); }
    s += '}';
    mark("proof text");
    char *buf = (char *)malloc(s.size() + 1);
    if (!buf) { ctx->err = "out of host memory for the proof text"; return ZP_ERR_NOMEM; }
    memcpy(buf, s.data(), s.size() + 1);
    *out_json = buf;
    *out_len = s.size();
    return ZP_OK;
}

// no C++ exception may cross the C ABI (a Rust or ctypes caller cannot unwind it): allocation failures of the host-side vectors /
// strings (sizes follow caller parameters: n_queries * path words, 3 << fri_final_log, ...) come back as error codes
static int32_t prove_guarded(zp_ctx *ctx, bool bn, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                             size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf, int32_t fri_final_log,
                             int32_t n_queries, int32_t pow_bits, char **out_json, size_t *out_len) {
    if (!ctx) return ZP_ERR_ARG;
    if (out_json) *out_json = nullptr;
    if (out_len) *out_len = 0;
    try {
        return prove_impl(ctx, bn, air_name, h_program, program_words, d_trace, trace_words, h_pubs, n_pubs, logn, logb, fri_logf, fri_final_log,
                          n_queries, pow_bits, out_json, out_len);
    } catch (const std::bad_alloc &) {
        try { ctx->err = "out of host memory while building the proof"; } catch (...) {}
        return ZP_ERR_NOMEM;
    } catch (const std::exception &e) {
        try { ctx->err = std::string("internal error: ") + e.what(); } catch (...) {}
        return ZP_ERR_INTERNAL;
    } catch (...) {
        return ZP_ERR_INTERNAL;
    }
}

int32_t zp_stark_prove(zp_ctx *ctx, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                       size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf,
                       int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, char **out_json, size_t *out_len) {
    return prove_guarded(ctx, false, air_name, h_program, program_words, d_trace, trace_words, h_pubs, n_pubs, logn, logb, fri_logf, fri_final_log,
                         n_queries, pow_bits, out_json, out_len);
}

// the same prover in BN128-hash mode (the last STARK before the Groth16 wrap): 16-ary Poseidon-BN254 trees, transcript over the
// BN254 scalar field, no grinding.  zp_set_poseidon_bn254(ctx, 17, ...) must have installed the tables.
int32_t zp_stark_prove_bn128(zp_ctx *ctx, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                             size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf,
                             int32_t fri_final_log, int32_t n_queries, char **out_json, size_t *out_len) {
    return prove_guarded(ctx, true, air_name, h_program, program_words, d_trace, trace_words, h_pubs, n_pubs, logn, logb, fri_logf, fri_final_log,
                         n_queries, 0, out_json, out_len);
}

// Binary openings of the LAST proof zp_stark_prove_bn128 made on this ctx (what its text carries under "roots", "fri.roots" and "queries"):
//   [0] "PZOPEN01" [1] n_queries [2] n_trees [3] log2 of the LDE size (bits of a query index), then per tree (trace, quotient, [stage 2], FRI
//   layers): values per leaf, leaves, levels of the 16-ary tree; per tree the root (4 words); per query: the index, then per tree values[width]
//   and path[levels][16][4].  *out points into the ctx (valid until the next proof on it); ZP_ERR_ARG when there is none.
int32_t zp_stark_openings(zp_ctx *ctx, const uint64_t **out, size_t *words) {
    if (!ctx || !out || !words) return ZP_ERR_ARG;
    ZP_ARG(ctx, !ctx->last_openings.empty(), "no BN128-mode proof has been made on this ctx");
    *out = (const uint64_t *)ctx->last_openings.data();
    *words = ctx->last_openings.size();
    return ZP_OK;
}

int32_t zp_sha256(const uint8_t *data, size_t len, uint8_t *out32) {
    if ((!data && len) || !out32) return ZP_ERR_ARG;
    Sha256::digest(data, len, out32);
    return ZP_OK;
}

}  // extern "C"

// ======================================================================================================================
// zp_stark_prove_sharded: ONE chunk STARK over the G ranks of a communicator (BASELINE configs[3]) -- what
// eigen_zeth_amd/stark/sharded.py orchestrates in Python, behind one C-ABI call per rank, so that a compiled host (Rust: the side of
// src/prover/provider.rs:358-377) can spread one GenChunkProof over the GPUs of a node.  Every rank passes ITS W/G trace columns and
// ends with the same proof text, byte for byte the text zp_stark_prove writes for the whole trace on one GPU.
//     trace LDE                  columns [W/G][N] -> [W/G][M]                         no exchange
//     trace commitment           rows [W][M/G]: local subtree                          ONE all-to-all, all-gather of G sub-roots
//     stage-2 columns            replicated (a handful of columns)                     broadcast of the witness columns they read
//     constraint quotient        rows, with a blow-up halo (b rows of the next rank)   all-gather of G x Wt x b halo values
//     quotient commitment        rows: local subtree                                   all-gather of the quotient rows, of G sub-roots
//     out-of-domain evaluations  columns (coefficients never move)                     all-gather of the evaluations
//     DEEP quotient              rows                                                  all-gather of the result: "gather before FRI"
//     FRI, proof of work         replicated                                            none
//     query openings             the owner of a row answers                            one all-reduce of values and sub-tree paths
namespace {

struct ShardTop {                       // the top log2 G levels of a row-sharded tree, known to every rank
    std::vector<std::vector<u64>> levels;   // levels[l]: (G >> l) nodes of 4 words
    u64 root[4];
};

int32_t shard_top(zp_ctx *ctx, const std::vector<u64> &subroots, int G, ShardTop *out) {
    out->levels.clear();
    std::vector<u64> lvl = subroots;
    void *d = nullptr;
    if (G > 1) PV_TRY(zp_dev_alloc(ctx, (size_t)(G / 2) * 12 * 8, &d));
    int32_t rc = ZP_OK;
    for (int n = G; rc == ZP_OK && n > 1; n >>= 1) {
        out->levels.push_back(lvl);
        std::vector<u64> st((size_t)(n / 2) * 12, 0);
        for (int i = 0; i < n / 2; i++) memcpy(&st[12 * (size_t)i], &lvl[8 * (size_t)i], 64);
        rc = zp_h2d(ctx, d, st.data(), st.size() * 8);
        if (rc == ZP_OK) rc = zp_poseidon_perm(ctx, (uint64_t *)d, (size_t)(n / 2));
        if (rc == ZP_OK) rc = zp_d2h(ctx, st.data(), d, st.size() * 8);
        lvl.assign((size_t)(n / 2) * 4, 0);
        for (int i = 0; i < n / 2; i++) memcpy(&lvl[4 * (size_t)i], &st[12 * (size_t)i], 32);
    }
    if (d) (void)zp_dev_free(ctx, d);
    if (rc == ZP_OK) memcpy(out->root, lvl.data(), 32);
    return rc;
}

// local subtree over `rows` u64[Wc][nloc] + all-gather of the sub-roots + the top of the tree
int32_t shard_commit(zp_comm *comm, zp_ctx *ctx, DevBufs &dev, const u64 *rows, size_t nloc, int Wc, int G, u64 **tree_out, ShardTop *top) {
    u64 *tree, *sub;
    PV_TRY(dev.alloc((2 * nloc - 1) * 4, &tree));
    PV_TRY(dev.alloc((size_t)G * 4, &sub));
    PV_TRY(zp_merkle_commit(ctx, (const uint64_t *)rows, nloc, Wc, (uint64_t *)tree));
    PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)(tree + (2 * nloc - 2) * 4), (uint64_t *)sub, 4));
    std::vector<u64> h((size_t)G * 4);
    PV_TRY(zp_d2h(ctx, h.data(), sub, h.size() * 8));
    dev.release(sub);
    PV_TRY(shard_top(ctx, h, G, top));
    *tree_out = tree;
    return ZP_OK;
}

// BN128 mode (16-ary Poseidon-BN254 trees), one row per leaf: the local tree over my nloc = 16^h c rows holds the global tree's nodes of
// levels 0..h under my rows (a group of 16 nodes of level k < h covers 16^(k+1) aligned rows: inside one shard); the c nodes of level h go
// through ONE all-gather and every rank finishes the few levels above them (`top`: a 16-ary tree whose "leaves" are the G c digests)
struct ShardTopBn {
    u64 *ltree = nullptr, *top = nullptr;    // device: local tree (zp_merkle16_nodes(nloc) nodes), top tree (zp_merkle16_nodes(G c) nodes)
    size_t h = 0, ntop = 0;                  // local levels that are global levels; nodes of level h in the whole tree
    u64 root[4];
};
int32_t shard_commit_bn(zp_comm *comm, zp_ctx *ctx, DevBufs &dev, const u64 *rows, size_t nloc, int Wc, int G, ShardTopBn *out) {
    size_t h = 0, c = nloc, off = 0;
    while (c >= 16 && c % 16 == 0) { off += c; c /= 16; h++; }
    out->h = h;
    out->ntop = (size_t)G * c;
    PV_TRY(dev.alloc(zp_merkle16_nodes(nloc) * 4, &out->ltree));
    PV_TRY(dev.alloc(zp_merkle16_nodes(out->ntop) * 4, &out->top));
    PV_TRY(zp_merkle16_commit_bn254(ctx, (const uint64_t *)rows, nloc, Wc, (uint64_t *)out->ltree));
    PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)(out->ltree + off * 4), (uint64_t *)out->top, c * 4));
    PV_TRY(zpi_merkle16_levels_bn254(ctx, out->top, out->ntop));
    return zp_d2h(ctx, out->root, out->top + (zp_merkle16_nodes(out->ntop) - 1) * 4, 32);
}

// [G][C][nloc] (what an all-gather of [C][nloc] row shards delivers) -> [C][G * nloc]
int32_t shard_join(zp_ctx *ctx, const u64 *gathered, u64 *full, int C, size_t nloc, int G) {
    for (int h = 0; h < G; h++)
        ZP_HIP(ctx, hipMemcpy2DAsync(full + (size_t)h * nloc, (size_t)G * nloc * 8, gathered + (size_t)h * C * nloc, nloc * 8, nloc * 8, (size_t)C,
                                     hipMemcpyDeviceToDevice, ctx->stream));
    return ZP_OK;
}

int32_t prove_sharded_impl(zp_comm *comm, zp_ctx *ctx, const char *air_name, const uint64_t *h_program, size_t program_words,
                           const uint64_t *d_trace, size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb,
                           int32_t fri_logf, int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, bool bn, char **out_json, size_t *out_len) {
    ZpStage stage_(ctx, bn ? "stark_prove_sharded_bn128" : "stark_prove_sharded");
    const int G = zp_comm_world(comm), rank = zp_comm_rank(comm);
    ZP_ARG(ctx, air_name && h_program && out_json && out_len && (h_pubs || n_pubs == 0), "null pointer");
    {
        const size_t nl = strlen(air_name);
        bool ok = nl >= 1 && nl <= 64;
        for (size_t i = 0; ok && i < nl; i++) {
            const char ch = air_name[i];
            ok = (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z') || (ch >= '0' && ch <= '9') || ch == '_' || ch == '-' || ch == '.';
        }
        ZP_ARG(ctx, ok, "air_name must be 1..64 characters of [A-Za-z0-9_.-]");
    }
    ZP_ARG(ctx, program_words >= 12, "constraint program shorter than its header");
    static const unsigned char magic[8] = {'Z', 'P', 'A', 'I', 'R', '1', 0, 0};
    ZP_ARG(ctx, memcmp(h_program, magic, 8) == 0, "not a ZPAIR1 constraint program");
    const size_t W = h_program[1], W2 = h_program[2], n_pub_prog = h_program[4], n_chal = h_program[5], n_const = h_program[6],
                 n_instr = h_program[7], K = h_program[8], n_s2 = h_program[10], Q = h_program[11];
    std::vector<ZpFixedCol> fxc;
    ZP_ARG(ctx, n_const < (1u << 16) && n_instr < (1u << 24) && n_s2 < (1u << 16) && zpi_program_fixed_table(h_program, program_words, &fxc),
           "constraint program length does not match its header");
    ZP_ARG(ctx, (size_t)n_pubs == n_pub_prog, "number of public inputs does not match the program");
    ZP_ARG(ctx, W >= 1 && W < 4096 && W2 < 4096 && K >= 1 && Q >= 1 && Q <= 16, "program dimensions out of range");
    ZP_ARG(ctx, logn >= 1 && logb >= 1 && logn + logb <= 30 && fri_logf >= 1 && fri_logf <= 4 && fri_final_log >= 0 && fri_final_log < logn &&
                    n_queries >= 1 && n_queries <= 4096 && pow_bits >= 0 && pow_bits <= 40, "STARK parameters out of range");
    ZP_ARG(ctx, Q <= ((size_t)1 << logb), "the blow-up must cover the quotient degree");
    ZP_ARG(ctx, (n_s2 == 0) == (W2 == 0) && (n_s2 == 0 || n_chal == 3), "stage-2 table and widths disagree");
    for (int i = 0; i < n_pubs; i++) ZP_ARG(ctx, h_pubs[i] < GL_P, "public input not canonical");
    const u64 *stage2 = (const u64 *)h_program + 12 + n_const + n_instr;
    size_t w2sum = 0;
    for (size_t k = 0; k < n_s2; k++) {
        const u64 kind = stage2[4 * k];
        ZP_ARG(ctx, kind == 1 || kind == 2, "unknown stage-2 argument");
        ZP_ARG(ctx, stage2[4 * k + 1] < W && stage2[4 * k + 2] < W && stage2[4 * k + 3] < W, "stage-2 column out of range");
        w2sum += kind == 1 ? 3 : 9;
    }
    ZP_ARG(ctx, w2sum == W2, "stage-2 width does not match its table");
    const int logm = logn + logb;
    const size_t N = (size_t)1 << logn, M = (size_t)1 << logm, Wt = W + W2, b = (size_t)1 << logb;
    ZP_ARG(ctx, G >= 1 && M % (size_t)G == 0, "the domain rows must split evenly over the ranks");
    // columns: rank r owns [r wl, min((r + 1) wl, W)), wl = ceil(W / G) -- the last ranks may hold fewer (a 47-column verifier AIR over
    // 8 ranks: 6,6,6,6,6,6,6,5) or none; the exchange moves wl columns per rank, the missing ones as zeros (they land behind column W)
    const size_t wl = (W + (size_t)G - 1) / (size_t)G, nloc = M / G, r0 = (size_t)rank * nloc;
    const size_t wr = (size_t)rank * wl >= W ? 0 : (W - (size_t)rank * wl < wl ? W - (size_t)rank * wl : wl);      // my real columns
    ZP_ARG(ctx, d_trace || wr == 0, "null pointer");
    ZP_ARG(ctx, nloc >= b && nloc % b == 0 && nloc >= 2, "row shards must hold whole blow-up groups");
    ZP_ARG(ctx, trace_words == (wr << logn), "trace_words must be (this rank's columns) * 2^logn: ceil(W / world) columns per rank, the tail ranks fewer");
    const u64 shift = ctx->coset_shift, root32 = ctx->root32;
    const u64 wN = gl_root(root32, logn);
    DevBufs dev(ctx);

    uint8_t dg[32];
    program_digest(ctx, h_program, program_words, dg);
    char dg_hex[17];
    for (int i = 0; i < 8; i++) snprintf(dg_hex + 2 * i, 3, "%02x", dg[i]);
    std::vector<u64> first = {(u64)logn, (u64)logb, (u64)W, (u64)W2, (u64)fri_logf, (u64)fri_final_log, (u64)n_queries, (u64)pow_bits, root32, shift};
    for (int i = 0; i < 4; i++) {
        u64 wd = 0;
        for (int k = 0; k < 8; k++) wd |= (u64)dg[8 * i + k] << (8 * k);
        first.push_back(wd % GL_P);
    }
    first.push_back((u64)n_pubs);
    Transcript tr(ctx, bn);
    const Trees T{ctx, bn};
    if (n_pubs <= 64) {
        for (int i = 0; i < n_pubs; i++) first.push_back(h_pubs[i]);
        tr.absorb(first);
    } else {            // long public vectors enter through their commitment (replicated: a few thousand permutations at most)
        tr.absorb(first);
        size_t Mp;
        std::vector<u64> mat;
        if (!bn) {
            Mp = 2;
            while (Mp * 8 < (size_t)n_pubs) Mp <<= 1;
            mat.assign(Mp * 8, 0);
            for (int i = 0; i < n_pubs; i++) mat[i] = h_pubs[i];
        } else {        // BN128 mode: rows of 48 values, column-major, 16-ary tree (as in prove_impl)
            Mp = ((size_t)n_pubs + 47) / 48;
            mat.assign(Mp * 48, 0);
            for (int i = 0; i < n_pubs; i++) mat[(size_t)(i % 48) * Mp + (size_t)(i / 48)] = h_pubs[i];
        }
        u64 *dmat, *dtree;
        PV_TRY(dev.alloc(mat.size(), &dmat));
        PV_TRY(dev.alloc(T.tree_words(Mp), &dtree));
        PV_TRY(zp_h2d(ctx, dmat, mat.data(), mat.size() * 8));
        if (!bn) PV_TRY(zp_merkle_commit_rows(ctx, (const uint64_t *)dmat, Mp, 8, (uint64_t *)dtree));
        else PV_TRY(T.commit(dmat, Mp, 48, dtree));
        u64 rootp[4];
        PV_TRY(T.root(dtree, Mp, rootp));
        dev.release(dmat);
        dev.release(dtree);
        tr.absorb_root(rootp);
    }

    // 1. trace: LDE of my columns, ONE exchange columns -> rows, local subtree, sub-roots
    u64 *ext, *tree1 = nullptr;
    ShardTop top1, top2, topq;
    // BN128 mode (the last STARK before the Groth16 wrap, zp_stark_prove_bn128's text): a leaf of a narrow tree holds 2^g rows i, i + M', ...
    // (prove_impl: rows_per_leaf_log) -- rows of DIFFERENT shards.  The trace tree is the sharded one and must have one row per leaf
    // (W > 28: the 47-column verifier AIR this mode exists for); the stage-2 and quotient trees are narrow, their columns are whole on
    // every rank anyway (stage 2 is replicated, the quotient is gathered for its out-of-domain evaluation), and 2^g rows per leaf make
    // them M / 2^g permutations against the trace tree's M: they are committed and opened replicated, exactly as prove_impl does
    auto rows_per_leaf_log = [&](size_t width) {
        int g = 0;
        if (bn && width)
            while ((width << (g + 1)) <= 56 && g + 1 <= logm - 4) g++;
        return g;
    };
    ZP_ARG(ctx, !bn || rows_per_leaf_log(W) == 0, "BN128 mode shards the trace tree by rows: it needs one row per leaf (more than 28 trace columns)");
    ShardTopBn bt1;
    u64 root1[4], root2[4] = {0, 0, 0, 0}, rootq[4];
    PV_TRY(dev.alloc((Wt > (size_t)G * wl ? Wt : (size_t)G * wl) * nloc, &ext));   // [Wt][nloc]: ALL columns (trace, then stage 2), my rows (room for the exchange's zero columns)
    // (no coefficient buffer since round 5: my columns' out-of-domain evaluations come from d_trace itself, zp_ood_eval on the trace domain)
    {
        u64 *extc, *pack;
        PV_TRY(dev.alloc(wl * M, &extc));
        PV_TRY(dev.alloc(wl * M, &pack));
        if (wr < wl) PV_TRY(zp_dev_zero(ctx, extc + wr * M, (wl - wr) * M * 8));
        if (wr) PV_TRY(zp_lde(ctx, d_trace, (uint64_t *)extc, nullptr, logn, logb, (int32_t)wr, shift));
        PV_TRY(zp_exchange_columns_to_rows(comm, (const uint64_t *)extc, wl, M, (uint64_t *)pack, (uint64_t *)ext));
        PV_TRY(zp_sync(ctx));
        dev.release(extc);
        dev.release(pack);
    }
    if (bn) {
        PV_TRY(shard_commit_bn(comm, ctx, dev, ext, nloc, (int)W, G, &bt1));
        memcpy(root1, bt1.root, 32);
    } else {
        PV_TRY(shard_commit(comm, ctx, dev, ext, nloc, (int)W, G, &tree1, &top1));
        memcpy(root1, top1.root, 32);
    }
    tr.absorb_root(root1);
    std::vector<u64> pubchal(h_pubs, h_pubs + n_pubs);
    const int g2 = rows_per_leaf_log(W2);
    const size_t M2 = M >> g2, W2g = W2 << g2;
    u64 *ext2_kept = nullptr;                     // BN128 mode: the stage-2 extension, whole (its tree's leaves mix rows of all shards)
    u64 *tree2 = nullptr, *s2 = nullptr;          // s2: the stage-2 columns on the trace domain (replicated), kept for their out-of-domain evaluations
    if (n_s2) {
        const e3 chal = tr.challenge();
        PV_TRY(tr.rc);
        u64 *colb, *ext2;
        PV_TRY(dev.alloc(W2 * N, &s2));
        PV_TRY(dev.alloc(3 * N, &colb));           // the (at most three) witness columns an argument reads, on every rank
        auto column = [&](u64 idx, u64 *dst) -> int32_t {       // broadcast from the rank that owns it
            const int owner = (int)(idx / wl);
            if (owner == rank) PV_TRY(zp_d2d(ctx, dst, d_trace + (idx % wl) * N, N * 8));
            return zp_comm_broadcast(comm, (uint64_t *)dst, N, owner);
        };
        size_t at = 0;
        for (size_t k = 0; k < n_s2; k++) {
            const u64 *st = stage2 + 4 * k;
            PV_TRY(column(st[1], colb));
            PV_TRY(column(st[2], colb + N));
            if (st[0] == 1) {
                PV_TRY(zp_grand_product(ctx, (const uint64_t *)colb, (const uint64_t *)(colb + N), N, (const uint64_t *)chal.c, (uint64_t *)(s2 + at * N)));
                at += 3;
            } else {
                PV_TRY(column(st[3], colb + 2 * N));
                PV_TRY(zp_logup_columns(ctx, (const uint64_t *)colb, (const uint64_t *)(colb + N), (const uint64_t *)(colb + 2 * N), N, (const uint64_t *)chal.c,
                                        (uint64_t *)(s2 + at * N)));
                at += 9;
            }
        }
        PV_TRY(dev.alloc(W2 * M, &ext2));
        PV_TRY(zp_lde(ctx, (const uint64_t *)s2, (uint64_t *)ext2, nullptr, logn, logb, (int32_t)W2, shift));   // replicated: W2 << W
        ZP_HIP(ctx, hipMemcpy2DAsync(ext + W * nloc, nloc * 8, ext2 + r0, M * 8, nloc * 8, W2, hipMemcpyDeviceToDevice, ctx->stream));
        PV_TRY(zp_sync(ctx));
        dev.release(colb);
        if (bn) {
            ext2_kept = ext2;
            PV_TRY(dev.alloc(T.tree_words(M2), &tree2));
            PV_TRY(T.commit(ext2, M2, (int)W2g, tree2));
            PV_TRY(T.root(tree2, M2, root2));
        } else {
            dev.release(ext2);
            PV_TRY(shard_commit(comm, ctx, dev, ext + W * nloc, nloc, (int)W2, G, &tree2, &top2));
            memcpy(root2, top2.root, 32);
        }
        tr.absorb_root(root2);
        for (int c = 0; c < 3; c++) pubchal.push_back(chal.c[c]);
    }
    const e3 alpha = tr.challenge();
    PV_TRY(tr.rc);

    // 2. constraint quotient on my rows (+ the b halo rows of the next rank)
    u64 *dq_l;
    {
        const size_t fwords = zp_fixed_columns_words(h_program, program_words, logn, logb);
        u64 *fixed, *fx_l;             // (round 5: from the ctx's cache, as in prove_impl -- they were rebuilt for every proof)
        PV_TRY(fixed_columns_cached(ctx, h_program, program_words, fxc, h_pubs, n_pubs, logn, logb, shift, root32, dg_hex, &fixed));
        // my window of the two selectors, then the (whole, periodic) extra columns: the layout zp_eval_quotient_rows reads
        PV_TRY(dev.alloc(2 * nloc + (fwords - 2 * M), &fx_l));
        PV_TRY(zp_d2d(ctx, fx_l, fixed + r0, nloc * 8));
        PV_TRY(zp_d2d(ctx, fx_l + nloc, fixed + M + r0, nloc * 8));
        if (fwords > 2 * M) PV_TRY(zp_d2d(ctx, fx_l + 2 * nloc, fixed + 2 * M, (fwords - 2 * M) * 8));
        std::vector<u64> apow(3 * K);
        {
            e3 cur = e3_make(1, 0, 0);
            for (size_t k = 0; k < K; k++) { memcpy(&apow[3 * k], cur.c, 24); cur = e3_mul(cur, alpha); }
        }
        std::vector<u64> zhinv(b);
        {
            const u64 sN = gl_pow(shift, (u64)N), wb = gl_root(root32, logb);
            u64 p = 1;
            for (size_t j = 0; j < b; j++) { zhinv[j] = gl_inv(gl_sub(gl_mul(sN, p), 1)); p = gl_mul(p, wb); }
        }
        PV_TRY(dev.alloc(3 * nloc, &dq_l));
        // the generated kernel of this program in its row-window form, when the host registered one (zp_stark_set_air_kernel_rows): the small
        // operands go up in one buffer [pub | 0 | apow | zhinv], as in prove_impl
        zp_air_quotient_rows_fn plug = nullptr;
        u64 *d_ops = nullptr;
        size_t o_ap = 0, o_zh = 0;
        const uint64_t *xlo = nullptr, *xhi = nullptr;
        int32_t xlb = 0;
        if (!ctx->air_kernels.empty()) {
            char hex[65];
            for (int i = 0; i < 32; i++) snprintf(hex + 2 * i, 3, "%02x", dg[i]);
            auto it = ctx->air_kernels.find(std::string("rows:") + std::string(hex, 64));
            if (it != ctx->air_kernels.end()) plug = (zp_air_quotient_rows_fn)it->second;
        }
        if (plug) {
            std::vector<u64> ops(pubchal);
            ops.push_back(0);
            o_ap = ops.size();
            ops.insert(ops.end(), apow.begin(), apow.end());
            o_zh = ops.size();
            ops.insert(ops.end(), zhinv.begin(), zhinv.end());
            PV_TRY(dev.alloc(ops.size(), &d_ops));
            PV_TRY(zp_h2d(ctx, d_ops, ops.data(), ops.size() * 8));
            PV_TRY(zp_domain_tables(ctx, logm, &xlo, &xhi, &xlb));
        }
        auto run_plug = [&](const u64 *cols, size_t sc, size_t row0, size_t nrows) -> int32_t {
            const int hrc = plug((void *)ctx->stream, cols, (u64)sc, (const u64 *)fx_l, (u64)nloc, (u64)M, (u64)b, (u64)row0, (u64)nrows, d_ops, d_ops + o_ap,
                                 d_ops + o_zh, (const u64 *)xlo, (const u64 *)xhi, (int)xlb, shift, gl_inv(wN), dq_l, (u64)nloc);
            if (hrc != 0) {
                ctx->err = "generated constraint kernel (row window): launch failed (hip error " + std::to_string(hrc) + ")";
                return ZP_ERR_HIP;
            }
            return ZP_OK;
        };
        if (G == 1 && plug) {
            PV_TRY(run_plug((const u64 *)ext, nloc, 0, M));
        } else if (G == 1) {
            PV_TRY(zp_eval_quotient_rows(ctx, h_program, program_words, (const uint64_t *)ext, nloc, (const uint64_t *)fx_l, nloc, logm, logb, 0, M,
                                         (const uint64_t *)pubchal.data(), (int32_t)pubchal.size(), (const uint64_t *)apow.data(),
                                         (const uint64_t *)zhinv.data(), shift, gl_inv(wN), (uint64_t *)dq_l, nloc));
        } else {
            u64 *heads, *allh, *buf;
            PV_TRY(dev.alloc(Wt * b, &heads));
            PV_TRY(dev.alloc((size_t)G * Wt * b, &allh));
            PV_TRY(dev.alloc(Wt * (nloc + b), &buf));
            ZP_HIP(ctx, hipMemcpy2DAsync(heads, b * 8, ext, nloc * 8, b * 8, Wt, hipMemcpyDeviceToDevice, ctx->stream));     // my first b rows
            PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)heads, (uint64_t *)allh, Wt * b));
            ZP_HIP(ctx, hipMemcpy2DAsync(buf, (nloc + b) * 8, ext, nloc * 8, nloc * 8, Wt, hipMemcpyDeviceToDevice, ctx->stream));
            ZP_HIP(ctx, hipMemcpy2DAsync(buf + nloc, (nloc + b) * 8, allh + (size_t)((rank + 1) % G) * Wt * b, b * 8, b * 8, Wt, hipMemcpyDeviceToDevice,
                                         ctx->stream));                                                                 // ... of the next rank
            if (plug) {
                PV_TRY(run_plug((const u64 *)buf, nloc + b, r0, nloc));
            } else {
                PV_TRY(zp_eval_quotient_rows(ctx, h_program, program_words, (const uint64_t *)buf, nloc + b, (const uint64_t *)fx_l, nloc, logm, logb, r0,
                                             nloc, (const uint64_t *)pubchal.data(), (int32_t)pubchal.size(), (const uint64_t *)apow.data(),
                                             (const uint64_t *)zhinv.data(), shift, gl_inv(wN), (uint64_t *)dq_l, nloc));
            }
            PV_TRY(zp_sync(ctx));
            dev.release(heads);
            dev.release(allh);
            dev.release(buf);
        }
        PV_TRY(zp_sync(ctx));
        dev.release(fx_l);
        if (d_ops) dev.release(d_ops);
    }
    // the quotient is needed whole on every rank: for its out-of-domain evaluation and, with Q > 1, for the coefficients its pieces are slices of
    u64 *dq_whole, *dq_rows = dq_l, *treeq;       // dq_whole: u64[Wq][M], the committed quotient columns (all rows)
    int q_logn = logm;
    size_t Wq = 3;
    {
        u64 *gath;
        PV_TRY(dev.alloc((size_t)G * 3 * nloc, &gath));
        PV_TRY(dev.alloc(3 * M, &dq_whole));
        PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)dq_l, (uint64_t *)gath, 3 * nloc));
        PV_TRY(shard_join(ctx, gath, dq_whole, 3, nloc, G));
        PV_TRY(zp_sync(ctx));
        dev.release(gath);
    }
    if (Q > 1) {
        u64 *dqcoef, *pad, *pext;
        PV_TRY(dev.alloc(3 * M, &dqcoef));
        PV_TRY(zp_intt(ctx, (const uint64_t *)dq_whole, (uint64_t *)dqcoef, logm, 3));
        PV_TRY(dev.alloc(3 * Q * M, &pad));
        PV_TRY(zp_dev_zero(ctx, pad, 3 * Q * M * 8));
        for (size_t j = 0; j < Q; j++)
            for (int c = 0; c < 3; c++) PV_TRY(zp_d2d(ctx, pad + (3 * j + c) * M, dqcoef + c * M + j * N, N * 8));
        PV_TRY(dev.alloc(3 * Q * M, &pext));
        PV_TRY(zp_ntt(ctx, (const uint64_t *)pad, (uint64_t *)pext, logm, (int32_t)(3 * Q)));
        PV_TRY(dev.alloc(3 * Q * nloc, &dq_rows));
        ZP_HIP(ctx, hipMemcpy2DAsync(dq_rows, nloc * 8, pext + r0, M * 8, nloc * 8, 3 * Q, hipMemcpyDeviceToDevice, ctx->stream));   // my rows of the pieces
        PV_TRY(zp_sync(ctx));
        dev.release(pad);
        dev.release(dq_l);
        dev.release(dqcoef);
        dev.release(dq_whole);
        dq_whole = pext;
        q_logn = logn;
        Wq = 3 * Q;
    }
    const int qg = rows_per_leaf_log(Wq);
    const size_t Mq = M >> qg, Wqg = Wq << qg;
    if (bn) {
        PV_TRY(dev.alloc(T.tree_words(Mq), &treeq));
        PV_TRY(T.commit(dq_whole, Mq, (int)Wqg, treeq));
        PV_TRY(T.root(treeq, Mq, rootq));
    } else {
        PV_TRY(shard_commit(comm, ctx, dev, dq_rows, nloc, (int)Wq, G, &treeq, &topq));
        memcpy(rootq, topq.root, 32);
    }
    tr.absorb_root(rootq);
    const e3 zeta = tr.challenge();
    PV_TRY(tr.rc);

    // 3. out-of-domain evaluations (barycentric form, zp_ood_eval): every rank evaluates ITS trace columns from their values on the trace
    //    domain (d_trace: shift 1, stride 1), the evaluations are all-gathered; the replicated stage-2 columns and quotient on every rank
    const e3 zeta_w = e3_scale(zeta, wN);
    std::vector<u64> ev_all((Wt + Wq) * 3), ev_next(Wt * 3);
    {
        std::vector<u64> mine(2 * wl * 3, 0), all((size_t)G * 2 * wl * 3);
        if (wr) PV_TRY(zp_ood_eval(ctx, d_trace, N, 1, (int32_t)wr, logn, 1, (const uint64_t *)zeta.c, 1, (uint64_t *)mine.data(), (uint64_t *)(mine.data() + wl * 3)));
        u64 *dmine, *dall;
        PV_TRY(dev.alloc(mine.size(), &dmine));
        PV_TRY(dev.alloc(all.size(), &dall));
        PV_TRY(zp_h2d(ctx, dmine, mine.data(), mine.size() * 8));
        PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)dmine, (uint64_t *)dall, mine.size()));
        PV_TRY(zp_d2h(ctx, all.data(), dall, all.size() * 8));
        dev.release(dmine);
        dev.release(dall);
        for (int h = 0; h < G; h++) {
            const size_t c0 = (size_t)h * wl, have = c0 >= W ? 0 : (W - c0 < wl ? W - c0 : wl);
            if (!have) break;
            memcpy(&ev_all[c0 * 3], &all[(size_t)h * 2 * wl * 3], have * 24);
            memcpy(&ev_next[c0 * 3], &all[(size_t)h * 2 * wl * 3 + wl * 3], have * 24);
        }
    }
    if (W2) {
        PV_TRY(zp_ood_eval(ctx, (const uint64_t *)s2, N, 1, (int32_t)W2, logn, 1, (const uint64_t *)zeta.c, 1, (uint64_t *)(ev_all.data() + W * 3),
                           (uint64_t *)(ev_next.data() + W * 3)));
        PV_TRY(zp_sync(ctx));
        dev.release(s2);
    }
    PV_TRY(zp_ood_eval(ctx, (const uint64_t *)dq_whole, M, (size_t)1 << (logm - q_logn), (int32_t)Wq, q_logn, shift, (const uint64_t *)zeta.c, 0,
                       (uint64_t *)(ev_all.data() + Wt * 3), nullptr));
    PV_TRY(zp_sync(ctx));
    if (!bn) dev.release(dq_whole);               // BN128 mode opens its queries from the whole columns
    tr.absorb(ev_all);
    tr.absorb(ev_next);
    const e3 gamma = tr.challenge();
    PV_TRY(tr.rc);

    // 4. DEEP quotient on my rows, gathered before FRI
    u64 *df;
    {
        u64 *df_l, *gath;
        PV_TRY(dev.alloc(3 * nloc, &df_l));
        PV_TRY(dev.alloc((size_t)G * 3 * nloc, &gath));
        PV_TRY(dev.alloc(3 * M, &df));
        PV_TRY(zp_deep_quotient_rows(ctx, (const uint64_t *)ext, (int32_t)Wt, nloc, (const uint64_t *)dq_rows, (int32_t)Wq, nloc, logm, r0, nloc, (int32_t)Wt,
                                     (const uint64_t *)zeta.c, (const uint64_t *)zeta_w.c, (const uint64_t *)gamma.c, (const uint64_t *)ev_all.data(),
                                     (const uint64_t *)ev_next.data(), shift, (uint64_t *)df_l, nloc));
        PV_TRY(zp_comm_all_gather(comm, (const uint64_t *)df_l, (uint64_t *)gath, 3 * nloc));
        PV_TRY(shard_join(ctx, gath, df, 3, nloc, G));
        PV_TRY(zp_sync(ctx));
        dev.release(df_l);
        dev.release(gath);
    }

    // 5. FRI (replicated: a layer of the DEEP quotient is 3 columns)
    struct Layer { int lg, f; u64 *tree, *data; u64 root[4]; };
    std::vector<Layer> layers;
    int cur = logm;
    u64 cur_shift = shift;
    u64 *dlayer = df;
    while (cur > fri_final_log + logb) {
        const int f = fri_logf < cur - (fri_final_log + logb) ? fri_logf : cur - (fri_final_log + logb);
        Layer L;
        L.lg = cur; L.f = f; L.data = dlayer;
        const size_t m = (size_t)1 << (cur - f);
        PV_TRY(dev.alloc(T.tree_words(m), &L.tree));
        PV_TRY(T.commit(dlayer, m, 3 << f, L.tree));
        PV_TRY(T.root(L.tree, m, L.root));
        tr.absorb_root(L.root);
        const e3 beta = tr.challenge();
        PV_TRY(tr.rc);
        u64 *next;
        PV_TRY(dev.alloc((size_t)3 << (cur - f), &next));
        PV_TRY(zp_fri_fold(ctx, (const uint64_t *)dlayer, (uint64_t *)next, cur, f, (const uint64_t *)beta.c, cur_shift));
        layers.push_back(L);
        dlayer = next;
        cur_shift = gl_pow(cur_shift, (u64)1 << f);
        cur -= f;
    }
    const int final_log = cur;
    std::vector<u64> final_l((size_t)3 << final_log);
    PV_TRY(zp_d2h(ctx, final_l.data(), dlayer, final_l.size() * 8));
    for (int c = 0; c < 3; c++) tr.absorb(&final_l[(size_t)c << final_log], (size_t)1 << final_log);

    // 6. proof of work, then the queries: the owner of a row answers, one all-reduce spreads the answers
    u64 nonce = 0;
    if (pow_bits) {
        const std::vector<u64> seed = tr.squeeze(4);
        PV_TRY(tr.rc);
        PV_TRY(zp_pow_grind(ctx, (const uint64_t *)seed.data(), pow_bits, (uint64_t *)&nonce));
        tr.absorb(&nonce, 1);
    }
    std::vector<u64> qidx = tr.squeeze((size_t)n_queries);
    PV_TRY(tr.rc);
    for (u64 &v : qidx) v &= (M - 1);
    const size_t nq = (size_t)n_queries, depth = (size_t)logm;
    size_t dl = 0;
    while (((size_t)1 << dl) < nloc) dl++;
    // words of one path: Goldilocks mode `depth` digests; BN128 mode 16 digests per level of the 16-ary tree
    const size_t pwt = bn ? Trees::levels16(M) * 64 : depth * 4, pw2 = bn ? T.path_words(M2) : depth * 4, pwq = bn ? T.path_words(Mq) : depth * 4;
    const size_t ltw = bn ? bt1.h * 64 : dl * 4;          // ... of which the owner of the row supplies (the levels inside its sub-tree)
    std::vector<u64> v_tr(nq * W), p_tr(nq * pwt), v_s2(nq * W2g), p_s2(n_s2 ? nq * pw2 : 0), v_q(nq * Wqg), p_q(nq * pwq);
    {
        std::vector<int> own;
        std::vector<u64> lidx;
        for (size_t i = 0; i < nq; i++)
            if (qidx[i] >= r0 && qidx[i] < r0 + nloc) { own.push_back((int)i); lidx.push_back(qidx[i] - r0); }
        const size_t no = own.size();
        // layout of the reduced vector: per query [W | W2 | Wq values | local path words of each of the (2 or 3) trees]; BN128 mode: the trace
        // tree alone is sharded: [W values | local path words]
        const size_t ntree = bn ? 1 : n_s2 ? 3 : 2, nval = bn ? W : W + W2 + Wq, per = nval + ntree * ltw;
        const size_t lpw = bn ? Trees::levels16(nloc) * 64 : dl * 4;     // what the local tree's opening returns per query
        std::vector<u64> red(nq * per, 0), tv(no * (W > Wq ? (W > W2 ? W : W2) : (Wq > W2 ? Wq : W2))), tp(no * lpw);
        auto fill = [&](const u64 *mat, size_t Wc, size_t voff, const u64 *tree, size_t poff) -> int32_t {
            if (!no) return ZP_OK;
            PV_TRY(zp_gather_rows(ctx, (const uint64_t *)mat, nloc, (int32_t)Wc, (const uint64_t *)lidx.data(), (int32_t)no, (uint64_t *)tv.data()));
            if (bn) PV_TRY(zp_merkle16_open_batch_bn254(ctx, (const uint64_t *)tree, nloc, (const uint64_t *)lidx.data(), (int32_t)no, (uint64_t *)tp.data()));
            else PV_TRY(zp_merkle_open_batch(ctx, (const uint64_t *)tree, nloc, (const uint64_t *)lidx.data(), (int32_t)no, (uint64_t *)tp.data()));
            for (size_t k = 0; k < no; k++) {
                memcpy(&red[(size_t)own[k] * per + voff], &tv[k * Wc], Wc * 8);
                memcpy(&red[(size_t)own[k] * per + nval + poff], &tp[k * lpw], ltw * 8);     // the first bt1.h levels / all dl levels
            }
            return ZP_OK;
        };
        PV_TRY(fill(ext, W, 0, bn ? bt1.ltree : tree1, 0));
        if (!bn) {
            if (n_s2) PV_TRY(fill(ext + W * nloc, W2, W, tree2, ltw));
            PV_TRY(fill(dq_rows, Wq, W + W2, treeq, (ntree - 1) * ltw));
        }
        u64 *dred;
        PV_TRY(dev.alloc(red.size(), &dred));
        PV_TRY(zp_h2d(ctx, dred, red.data(), red.size() * 8));
        PV_TRY(zp_comm_all_reduce_sum(comm, (uint64_t *)dred, red.size()));
        PV_TRY(zp_d2h(ctx, red.data(), dred, red.size() * 8));
        dev.release(dred);
        auto paths = [&](std::vector<u64> &dst, size_t poff, const ShardTop &top) {
            for (size_t i = 0; i < nq; i++) {
                memcpy(&dst[i * pwt], &red[i * per + nval + poff], dl * 32);
                u64 node = qidx[i] >> dl;
                for (size_t l = 0; l < top.levels.size(); l++) {         // the top of the path comes from the all-gathered sub-roots
                    memcpy(&dst[i * pwt + (dl + l) * 4], &top.levels[l][(size_t)(node ^ 1) * 4], 32);
                    node >>= 1;
                }
            }
        };
        for (size_t i = 0; i < nq; i++) {
            memcpy(&v_tr[i * W], &red[i * per], W * 8);
            if (bn) continue;
            if (W2) memcpy(&v_s2[i * W2], &red[i * per + W], W2 * 8);
            memcpy(&v_q[i * Wq], &red[i * per + W + W2], Wq * 8);
        }
        if (!bn) {
            paths(p_tr, 0, top1);
            if (n_s2) paths(p_s2, ltw, top2);
            paths(p_q, (ntree - 1) * ltw, topq);
        } else {
            // trace: the levels above the shards from the replicated top tree (its "leaves" are the nodes of level h)
            const size_t tlv = Trees::levels16(bt1.ntop);
            ZP_ARG(ctx, bt1.h + tlv == Trees::levels16(M), "internal: levels of the sharded 16-ary tree do not add up");
            std::vector<u64> tidx(nq), topp(nq * tlv * 64);
            for (size_t i = 0; i < nq; i++) tidx[i] = qidx[i] >> (4 * bt1.h);
            PV_TRY(zp_merkle16_open_batch_bn254(ctx, (const uint64_t *)bt1.top, bt1.ntop, (const uint64_t *)tidx.data(), n_queries, (uint64_t *)topp.data()));
            for (size_t i = 0; i < nq; i++) {
                memcpy(&p_tr[i * pwt], &red[i * per + nval], ltw * 8);
                memcpy(&p_tr[i * pwt + ltw], &topp[i * tlv * 64], tlv * 64 * 8);
            }
            // stage 2 and quotient: replicated trees over whole columns, opened as prove_impl opens them
            std::vector<u64> rows(nq);
            if (n_s2) {
                for (size_t i = 0; i < nq; i++) rows[i] = qidx[i] & (M2 - 1);
                PV_TRY(zp_gather_rows(ctx, (const uint64_t *)ext2_kept, M2, (int32_t)W2g, (const uint64_t *)rows.data(), n_queries, (uint64_t *)v_s2.data()));
                PV_TRY(T.open(tree2, M2, rows.data(), n_queries, p_s2.data()));
            }
            for (size_t i = 0; i < nq; i++) rows[i] = qidx[i] & (Mq - 1);
            PV_TRY(zp_gather_rows(ctx, (const uint64_t *)dq_whole, Mq, (int32_t)Wqg, (const uint64_t *)rows.data(), n_queries, (uint64_t *)v_q.data()));
            PV_TRY(T.open(treeq, Mq, rows.data(), n_queries, p_q.data()));
        }
    }
    struct FriOpen { std::vector<u64> vals, paths; size_t width, depth, pw, m; };
    std::vector<FriOpen> fo(layers.size());
    {
        std::vector<u64> pos = qidx;
        for (size_t li = 0; li < layers.size(); li++) {
            const Layer &L = layers[li];
            const size_t m = (size_t)1 << (L.lg - L.f);
            for (u64 &p : pos) p &= (m - 1);
            fo[li].width = (size_t)3 << L.f;
            fo[li].depth = (size_t)(L.lg - L.f);
            fo[li].m = m;
            fo[li].pw = T.path_words(m);
            fo[li].vals.resize(nq * fo[li].width);
            fo[li].paths.resize(nq * fo[li].pw);
            PV_TRY(zp_gather_rows(ctx, (const uint64_t *)L.data, m, (int32_t)fo[li].width, (const uint64_t *)pos.data(), n_queries, (uint64_t *)fo[li].vals.data()));
            PV_TRY(T.open(L.tree, m, pos.data(), n_queries, fo[li].paths.data()));
        }
    }

    if (bn) {       // every rank keeps the binary openings record (zp_stark_openings), as zp_stark_prove_bn128 does
        std::vector<TreeOut> trees = {{W, M, root1, &v_tr, &p_tr, pwt}, {Wqg, Mq, rootq, &v_q, &p_q, pwq}};
        if (n_s2) trees.push_back({W2g, M2, root2, &v_s2, &p_s2, pw2});
        for (size_t li = 0; li < layers.size(); li++) trees.push_back({fo[li].width, fo[li].m, layers[li].root, &fo[li].vals, &fo[li].paths, fo[li].pw});
        openings_record(ctx, tr, qidx, logm, trees);
    }

    // the proof text: exactly what zp_stark_prove / zp_stark_prove_bn128 writes
    std::string s;
    s.reserve(nq * (Wt + Wq + 64) * 24 + (1 << 16));
    s += "{\"air\":\"";
    s += air_name;
    s += "\",\"air_digest\":\"";
    s += dg_hex;
    s += "\",\"params\":{\"logn\":";
    j_u64(s, (u64)logn); s += ",\"logb\":"; j_u64(s, (u64)logb); s += ",\"fri_logf\":"; j_u64(s, (u64)fri_logf);
    s += ",\"fri_final_log\":"; j_u64(s, (u64)fri_final_log); s += ",\"n_queries\":"; j_u64(s, (u64)n_queries);
    s += ",\"pow_bits\":"; j_u64(s, (u64)pow_bits);
    if (bn) s += ",\"hash\":\"bn128\"";
    s += "},\"root32\":"; j_u64(s, root32);
    s += ",\"shift\":"; j_u64(s, shift);
    s += ",\"publics\":"; j_list(s, (const u64 *)h_pubs, (size_t)n_pubs);
    s += ",\"roots\":{\"trace\":"; j_root(s, root1, bn);
    s += ",\"quotient\":"; j_root(s, rootq, bn);
    if (n_s2) { s += ",\"stage2\":"; j_root(s, root2, bn); }
    s += "},\"evals\":{\"z\":"; j_e3list(s, ev_all);
    s += ",\"zw\":"; j_e3list(s, ev_next);
    s += "},\"fri\":{\"roots\":[";
    for (size_t li = 0; li < layers.size(); li++) { if (li) s += ','; j_root(s, layers[li].root, bn); }
    s += "],\"final\":[";
    for (int c = 0; c < 3; c++) { if (c) s += ','; j_list(s, &final_l[(size_t)c << final_log], (size_t)1 << final_log); }
    s += "]},\"queries\":[";
    for (size_t i = 0; i < nq; i++) {
        if (i) s += ',';
        s += "{\"index\":"; j_u64(s, qidx[i]);
        auto opening = [&](const u64 *vals, size_t width, const u64 *path, size_t rows, size_t bin_depth) {
            if (bn) j_opening_bn(s, vals, width, path, Trees::levels16(rows));
            else j_opening(s, vals, width, path, bin_depth);
        };
        s += ",\"trace\":"; opening(&v_tr[i * W], W, &p_tr[i * pwt], M, depth);
        s += ",\"quotient\":"; opening(&v_q[i * Wqg], Wqg, &p_q[i * pwq], Mq, depth);
        if (n_s2) { s += ",\"stage2\":"; opening(&v_s2[i * W2g], W2g, &p_s2[i * pw2], M2, depth); }
        s += ",\"fri\":[";
        for (size_t li = 0; li < layers.size(); li++) {
            if (li) s += ',';
            opening(&fo[li].vals[i * fo[li].width], fo[li].width, &fo[li].paths[i * fo[li].pw], fo[li].m, fo[li].depth);
        }
        s += "]}";
    }
    s += ']';
    if (pow_bits) { s += ",\"pow_nonce\":"; j_u64(s, nonce); }
    s += '}';
    char *buf = (char *)malloc(s.size() + 1);
    if (!buf) { ctx->err = "out of host memory for the proof text"; return ZP_ERR_NOMEM; }
    memcpy(buf, s.data(), s.size() + 1);
    *out_json = buf;
    *out_len = s.size();
    return ZP_OK;
}

}  // namespace

namespace {
int32_t prove_sharded_guarded(zp_comm *comm, bool bn, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace_local,
                              size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf,
                              int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, char **out_json, size_t *out_len) {
    if (!comm) return ZP_ERR_ARG;
    zp_ctx *ctx = zpi_comm_ctx(comm);
    if (out_json) *out_json = nullptr;
    if (out_len) *out_len = 0;
    // Whatever stops this rank -- a failed allocation, a HIP error, an exception -- its peers are inside the same call, heading for
    // the next collective: zpi_comm_fail takes the communicator down so that they return ZP_ERR_COMM instead of waiting for ever.
    try {
        return zpi_comm_fail(comm, prove_sharded_impl(comm, ctx, air_name, h_program, program_words, d_trace_local, trace_words, h_pubs, n_pubs, logn, logb,
                                                      fri_logf, fri_final_log, n_queries, pow_bits, bn, out_json, out_len));
    } catch (const std::bad_alloc &) {
        try { ctx->err = "out of host memory while building the proof"; } catch (...) {}
        return zpi_comm_fail(comm, ZP_ERR_NOMEM);
    } catch (const std::exception &e) {
        try { ctx->err = std::string("internal error: ") + e.what(); } catch (...) {}
        return zpi_comm_fail(comm, ZP_ERR_INTERNAL);
    } catch (...) {
        return zpi_comm_fail(comm, ZP_ERR_INTERNAL);
    }
}
}  // namespace

extern "C" int32_t zp_stark_prove_sharded(zp_comm *comm, const char *air_name, const uint64_t *h_program, size_t program_words,
                                          const uint64_t *d_trace_local, size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn,
                                          int32_t logb, int32_t fri_logf, int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, char **out_json,
                                          size_t *out_len) {
    return prove_sharded_guarded(comm, false, air_name, h_program, program_words, d_trace_local, trace_words, h_pubs, n_pubs, logn, logb, fri_logf,
                                 fri_final_log, n_queries, pow_bits, out_json, out_len);
}

// the sharded prover in BN128-hash mode: zp_stark_prove_bn128's text (and openings record, on every rank) from W/world columns per rank.
// The trace tree is sharded by rows and must have one row per leaf (more than 28 trace columns: the verifier AIR of the final STARK);
// zp_set_poseidon_bn254(ctx, 17, ...) on every rank's ctx.
extern "C" int32_t zp_stark_prove_sharded_bn128(zp_comm *comm, const char *air_name, const uint64_t *h_program, size_t program_words,
                                                const uint64_t *d_trace_local, size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs,
                                                int32_t logn, int32_t logb, int32_t fri_logf, int32_t fri_final_log, int32_t n_queries,
                                                char **out_json, size_t *out_len) {
    return prove_sharded_guarded(comm, true, air_name, h_program, program_words, d_trace_local, trace_words, h_pubs, n_pubs, logn, logb, fri_logf,
                                 fri_final_log, n_queries, 0, out_json, out_len);
}
