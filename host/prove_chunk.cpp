// A compiled host above the C-ABI -- what a Rust prover service (the reference's side of the boundary is compiled code:
// src/prover/provider.rs talks to such a service) does for one GenChunkProof, written in C++ because this image has no Rust
// toolchain: load libzethprover.so's entry points, put the witness in HBM, call zp_stark_prove, hand the proof text on.
// Nothing but include/zeth_prover.h is used -- no Python, no torch, no compiler at run time (the AIR is a data blob).
//
// usage: prove_chunk <program.bin> <trace.bin> <publics.bin> <logn> <logb> <fri_logf> <fri_final_log> <n_queries> <pow_bits> <out.json> [air_name
//                     [rank world id-file [run-nonce]]]
//   with rank / world / id-file: ONE proof over `world` GPUs, one process per GPU (zp_stark_prove_sharded on an RCCL communicator; rank 0
//   publishes the 128-byte RCCL id in <id-file>, the others wait for the record that carries this run's nonce: host/rendezvous.hpp).  Rank r reads only ITS columns of trace.bin (ceil(W / world) per rank, the tail ranks fewer) and drives GPU r;
//   every rank obtains the same proof text (rank 0 writes <out.json>), byte for byte the single-GPU text.
//   program.bin : the constraint program blob (u64 words, layout in the header)
//   trace.bin   : u64[W][2^logn] column-major, canonical values
//   publics.bin : u64[n_pub]
// build: make -C host   (g++ -I../include prove_chunk.cpp -L../eigen_zeth_amd/csrc -lzethprover)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/zeth_prover.h"
#include "rendezvous.hpp"

static std::vector<uint64_t> read_words(const char *path) {
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    const long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> v((size_t)n / 8);
    if (n % 8 || fread(v.data(), 8, v.size(), f) != v.size()) { fprintf(stderr, "bad file %s\n", path); exit(2); }
    fclose(f);
    return v;
}

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int32_t rc_ = (call);                                                                   \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? zp_last_error(ctx) : "no context"); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 11) { fprintf(stderr, "usage: see the header of host/prove_chunk.cpp\n"); return 2; }
    const std::vector<uint64_t> program = read_words(argv[1]), trace = read_words(argv[2]), pubs = read_words(argv[3]);
    const int logn = atoi(argv[4]), logb = atoi(argv[5]), fri_logf = atoi(argv[6]), fri_final_log = atoi(argv[7]), n_queries = atoi(argv[8]),
              pow_bits = atoi(argv[9]);
    const char *air_name = argc > 11 ? argv[11] : "chunk";
    const bool sharded = argc > 14;
    const int rank = sharded ? atoi(argv[12]) : 0, world = sharded ? atoi(argv[13]) : 1;
    // the shapes must agree BEFORE anything reaches the GPU: W comes from the program header (word 1), the number of public
    // inputs from word 4; the library checks trace_words again (ZP_ERR_ARG otherwise)
    if (program.size() < 12) { fprintf(stderr, "%s: shorter than a constraint-program header\n", argv[1]); return 2; }
    if (program[1] < 1 || program[1] >= 4096) { fprintf(stderr, "%s: width %llu out of range\n", argv[1], (unsigned long long)program[1]); return 2; }   // before it is shifted
    if (logn < 1 || logn > 30 || trace.size() != (size_t)(program[1] << logn)) {
        fprintf(stderr, "%s: %zu words, the program needs W * 2^logn = %llu * 2^%d\n", argv[2], trace.size(), (unsigned long long)program[1], logn);
        return 2;
    }
    if (pubs.size() != program[4]) { fprintf(stderr, "%s: %zu public inputs, the program declares %llu\n", argv[3], pubs.size(), (unsigned long long)program[4]); return 2; }
    for (uint64_t v : trace)
        if (v >= 0xFFFFFFFF00000001ULL) { fprintf(stderr, "%s: non-canonical trace value (precondition of zp_stark_prove)\n", argv[2]); return 2; }
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad rank / world\n"); return 2; }
    zp_ctx *ctx = nullptr;
    CHECK(zp_create(&ctx, rank));                      // one process per GPU: rank r drives device r
    // rank r owns columns [r wl, min((r + 1) wl, W)), wl = ceil(W / world): the tail ranks hold fewer columns, or none (include/zeth_prover.h)
    const size_t Wc = (size_t)program[1], wl = (Wc + (size_t)world - 1) / (size_t)world, c0 = (size_t)rank * wl < Wc ? (size_t)rank * wl : Wc,
                 cols = Wc - c0 < wl ? Wc - c0 : wl, words = cols << logn;
    const uint64_t *mine = trace.data() + (c0 << logn);                // this rank's columns (a real host would read only these)
    void *d_trace = nullptr;
    CHECK(zp_dev_alloc(ctx, (words ? words : 1) * 8, &d_trace));
    if (words) CHECK(zp_h2d(ctx, d_trace, mine, words * 8));
    char *json = nullptr;
    size_t len = 0;
    zp_comm *comm = nullptr;
    if (sharded) {
        uint8_t id[128];
        const uint64_t nonce = argc > 15 ? strtoull(argv[15], nullptr, 0) : 0;
        if (rank == 0) {
            CHECK(zp_comm_unique_id(id));
            if (!zp_rendezvous::publish(argv[14], nonce, id)) { fprintf(stderr, "cannot write %s\n", argv[14]); return 2; }
        } else if (!zp_rendezvous::await(argv[14], nonce, id)) {
            fprintf(stderr, "no RCCL id of this run in %s after 60 s\n", argv[14]);
            return 2;
        }
        CHECK(zp_comm_create(ctx, rank, world, id, &comm));
        if (rank == 0) zp_rendezvous::retire(argv[14]);
        CHECK(zp_stark_prove_sharded(comm, air_name, program.data(), program.size(), (const uint64_t *)d_trace, words, pubs.data(), (int32_t)pubs.size(), logn,
                                     logb, fri_logf, fri_final_log, n_queries, pow_bits, &json, &len));
    } else {
        CHECK(zp_stark_prove(ctx, air_name, program.data(), program.size(), (const uint64_t *)d_trace, trace.size(), pubs.data(), (int32_t)pubs.size(), logn, logb,
                             fri_logf, fri_final_log, n_queries, pow_bits, &json, &len));
    }
    if (rank == 0) {
        FILE *o = fopen(argv[10], "wb");
        if (!o || fwrite(json, 1, len, o) != len) { fprintf(stderr, "cannot write %s\n", argv[10]); return 2; }
        fclose(o);
    }
    zp_free_buffer(json);
    if (comm) zp_comm_destroy(comm);
    zp_dev_free(ctx, d_trace);
    zp_destroy(ctx);
    printf("rank %d/%d proof: %zu bytes%s%s\n", rank, world, len, rank == 0 ? " -> " : "", rank == 0 ? argv[10] : "");
    return 0;
}
