// The out-of-band hand-over of the 128-byte RCCL id between the processes of one run, through a file (a Rust service would use its own
// channel; zp_comm_create only needs the bytes).  Safe against the two ways a file rendezvous goes wrong:
//   * a STALE file of an earlier run: the record carries a magic and the run's nonce (the launcher gives every rank of one run the same
//     nonce; 0 if it gives none), the readers take nothing else, rank 0 removes whatever is at the path before it writes and removes its own
//     record once the communicator stands (every rank has read it by then);
//   * a HALF-WRITTEN file: rank 0 writes to a temporary name and rename()s it into place (atomic on one file system).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <unistd.h>

namespace zp_rendezvous {

struct Record {
    char magic[8];
    uint64_t nonce;
    uint8_t id[128];
};

inline bool publish(const char *path, uint64_t nonce, const uint8_t id[128]) {      // rank 0
    Record r;
    memcpy(r.magic, "ZPRCCLID", 8);
    r.nonce = nonce;
    memcpy(r.id, id, 128);
    (void)unlink(path);
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((long)getpid());
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = fwrite(&r, 1, sizeof r, f) == sizeof r;
    if (fclose(f) != 0 || !ok) { (void)unlink(tmp.c_str()); return false; }
    if (rename(tmp.c_str(), path) != 0) { (void)unlink(tmp.c_str()); return false; }
    return true;
}

inline bool await(const char *path, uint64_t nonce, uint8_t id[128], int timeout_s = 60) {    // ranks > 0
    for (int tries = 0; tries < timeout_s * 10; tries++) {
        Record r;
        FILE *f = fopen(path, "rb");
        const bool got = f && fread(&r, 1, sizeof r, f) == sizeof r && memcmp(r.magic, "ZPRCCLID", 8) == 0 && r.nonce == nonce;
        if (f) fclose(f);
        if (got) { memcpy(id, r.id, 128); return true; }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
}

inline void retire(const char *path) { (void)unlink(path); }     // rank 0, after zp_comm_create returned (all ranks have joined)

}  // namespace zp_rendezvous
