// One trace commitment sharded over the GPUs of a node, from a compiled host on the C-ABI alone (no Python, no torch): what a Rust
// prover service does when it spreads ONE chunk proof's commitment stage over G GPUs (src/prover/provider.rs:358-377 sends one
// GenChunkProof; BASELINE.json configs[3] shards its columns 8-way).  One process per GPU:
//
//   commit_sharded <trace.bin> <logn> <logb> <W> <rank> <world> <id-file> [run-nonce]
//
// trace.bin: u64[W][2^logn] column-major (canonical).  Rank r owns columns [r W/G, (r+1) W/G): it extends them (zp_lde, no
// exchange), then zp_merkle_commit_sharded packs, runs the one RCCL all-to-all, hashes its M/G rows, all-gathers the sub-roots
// and finishes the tree.  Rank 0 publishes the 128-byte RCCL id in <id-file> (host/rendezvous.hpp: atomic, tagged with the run's
// nonce, removed once the communicator stands; the others wait for it); every rank prints the
// global root -- equal to the root a single GPU commits over all W columns.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/zeth_prover.h"
#include "rendezvous.hpp"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        const int32_t rc_ = (call);                                                                   \
        if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ctx ? zp_last_error(ctx) : "no context"); return 1; } \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 8) { fprintf(stderr, "usage: commit_sharded <trace.bin> <logn> <logb> <W> <rank> <world> <id-file> [run-nonce]\n"); return 2; }
    const int logn = atoi(argv[2]), logb = atoi(argv[3]), W = atoi(argv[4]), rank = atoi(argv[5]), world = atoi(argv[6]);
    if (logn < 1 || logn > 28 || logb < 0 || logb > 4 || W < 1 || world < 1 || rank < 0 || rank >= world || W % world) {
        fprintf(stderr, "bad arguments (W must be a multiple of world)\n");
        return 2;
    }
    const size_t N = (size_t)1 << logn, M = N << logb, Wl = (size_t)W / world;
    std::vector<uint64_t> cols(Wl * N);
    {
        FILE *f = fopen(argv[1], "rb");
        if (!f || fseek(f, (long)((size_t)rank * Wl * N * 8), SEEK_SET) != 0 || fread(cols.data(), 8, cols.size(), f) != cols.size()) {
            fprintf(stderr, "cannot read this rank's columns from %s\n", argv[1]);
            return 2;
        }
        fclose(f);
    }
    zp_ctx *ctx = nullptr;
    CHECK(zp_create(&ctx, rank));           // one process per GPU: rank r drives device r
    uint8_t id[128];
    const uint64_t nonce = argc > 8 ? strtoull(argv[8], nullptr, 0) : 0;
    if (rank == 0) {
        CHECK(zp_comm_unique_id(id));
        if (!zp_rendezvous::publish(argv[7], nonce, id)) { fprintf(stderr, "cannot write %s\n", argv[7]); return 2; }
    } else if (!zp_rendezvous::await(argv[7], nonce, id)) {
        fprintf(stderr, "no RCCL id of this run in %s after 60 s\n", argv[7]);
        return 2;
    }
    zp_comm *comm = nullptr;
    CHECK(zp_comm_create(ctx, rank, world, id, &comm));
    if (rank == 0) zp_rendezvous::retire(argv[7]);
    void *d_in = nullptr, *d_ext = nullptr, *d_tree = nullptr;
    CHECK(zp_dev_alloc(ctx, Wl * N * 8, &d_in));
    CHECK(zp_dev_alloc(ctx, Wl * M * 8, &d_ext));
    CHECK(zp_dev_alloc(ctx, (2 * (M / world) - 1) * 32, &d_tree));
    CHECK(zp_h2d(ctx, d_in, cols.data(), Wl * N * 8));
    CHECK(zp_lde(ctx, (const uint64_t *)d_in, (uint64_t *)d_ext, nullptr, logn, logb, (int32_t)Wl, 0));
    uint64_t root[4];
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(zp_merkle_commit_sharded(comm, (const uint64_t *)d_ext, M, (int32_t)Wl, (uint64_t *)d_tree, root));
    CHECK(zp_sync(ctx));
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    printf("rank %d/%d root %016llx %016llx %016llx %016llx  (sharded commit of %d x 2^%d: %.2f ms)\n", rank, world, (unsigned long long)root[0],
           (unsigned long long)root[1], (unsigned long long)root[2], (unsigned long long)root[3], W, logn + logb, ms);
    zp_comm_destroy(comm);
    zp_dev_free(ctx, d_in);
    zp_dev_free(ctx, d_ext);
    zp_dev_free(ctx, d_tree);
    zp_destroy(ctx);
    return 0;
}
