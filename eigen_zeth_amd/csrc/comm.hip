// Multi-GPU behind the C-ABI: RCCL over xGMI, called directly (BASELINE.json configs[3]).
//
// One process per GPU.  A compiled host (the reference's side of the boundary is Rust: src/prover/provider.rs:358-377 sends one
// GenChunkProof per batch) shards ONE commitment over the GPUs of a node without Python or torch: columns are owned by ranks
// (NTT / LDE need no exchange), leaf hashing needs whole rows, so there is exactly one exchange -- rank g sends rank h the
// rows [h M/G, (h+1) M/G) of its columns.  xGMI is point-to-point (7 links per GPU), so the exchange is a grouped
// ncclSend / ncclRecv all-to-all: every pairwise message rides its own link; the only other collective is the all-gather of
// G x 32 bytes of sub-roots.  librccl is loaded at run time (dlopen), so the library has no link-time dependency on it and a
// process that already carries an RCCL (a torch process) keeps using that one.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <functional>
#include <thread>
#include <vector>

#include "ctx.hpp"

namespace {

struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;                  // optional: the watchdog's way out of a collective a peer never joins
    decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr;  // optional
    decltype(&ncclCommCount) CommCount = nullptr;                  // optional: zp_comm_info asks the communicator itself how many ranks it joined
    decltype(&ncclCommUserRank) CommUserRank = nullptr;            // optional
    decltype(&ncclCommCuDevice) CommCuDevice = nullptr;            // optional
    bool ok = false;
};

Rccl &rccl() {
    static Rccl r;
    if (r.h || r.ok) return r;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names)          // an RCCL that is already in the process (same SONAME) is reused
        if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD))) break;
    if (!r.h)
        for (const char *n : names)
            if ((r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.h) return r;
#define ZP_SYM(name) r.name = (decltype(r.name))dlsym(r.h, "nccl" #name)
    ZP_SYM(GetUniqueId); ZP_SYM(CommInitRank); ZP_SYM(CommDestroy); ZP_SYM(GroupStart); ZP_SYM(GroupEnd);
    ZP_SYM(Send); ZP_SYM(Recv); ZP_SYM(AllGather); ZP_SYM(Broadcast); ZP_SYM(AllReduce); ZP_SYM(GetErrorString);
    ZP_SYM(CommAbort); ZP_SYM(CommGetAsyncError); ZP_SYM(CommCount); ZP_SYM(CommUserRank); ZP_SYM(CommCuDevice);
#undef ZP_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllGather && r.Broadcast &&
           r.AllReduce && r.GetErrorString;
    return r;
}

}  // namespace

// A communicator WITHOUT RCCL for ranks that live in one process (threads, one ctx each -- on one GPU or several): every collective is
// device-to-device copies between the ranks' buffers around a thread barrier.  RCCL refuses two ranks on one device, so this is how the
// multi-rank logic of the sharded entry points (zp_merkle_commit_sharded, zp_stark_prove_sharded) is exercised on a one-GPU box; it is
// also a legitimate transport for a single-process multi-GPU host (peer copies).
struct zp_comm_group {
    int world = 0;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned gen = 0;
    const void *ptr[64];
    // A rank that fails -- inside a collective or anywhere between two of them (zp_comm_abort) -- POISONS the group: every rank that is
    // waiting in a barrier wakes with ZP_ERR_COMM and every later collective returns it at once.  A rank that never arrives (a crashed
    // thread, a host that forgot the call) is caught by the barrier's timeout, which poisons the group the same way.  Nobody waits for ever.
    bool poisoned = false;
    int timeout_ms = 120000;
    void fail() {
        std::lock_guard<std::mutex> lk(mu);
        poisoned = true;
        cv.notify_all();
    }
    bool dead() {
        std::lock_guard<std::mutex> lk(mu);
        return poisoned;
    }
    int32_t barrier() {         // ZP_OK: every rank arrived; ZP_ERR_COMM: the group is dead (a peer failed or did not arrive in time)
        std::unique_lock<std::mutex> lk(mu);
        if (poisoned) return ZP_ERR_COMM;
        const unsigned g = gen;
        if (++waiting == world) {
            waiting = 0;
            gen++;
            cv.notify_all();
            return ZP_OK;
        }
        const bool woke = cv.wait_for(lk, std::chrono::milliseconds(timeout_ms), [&] { return gen != g || poisoned; });
        // The generation completed: every rank was counted and the last arriver has gone on to copy.  That is ZP_OK even when a peer has
        // poisoned the group SINCE (its copy failed while this waiter was still asleep): this rank must take part in the closing rendezvous
        // like everybody else -- returning ZP_ERR_COMM here made its caller free a published buffer that peers were still reading, and left
        // them waiting drain_ms for an arrival that never came.  closing_barrier() reports the poison.
        if (woke && gen != g) return ZP_OK;
        if (!poisoned) {        // timed out: take everybody else down too
            poisoned = true;
            cv.notify_all();
        }
        return ZP_ERR_COMM;
    }
    // The closing rendezvous of a collective.  Unlike barrier() it COUNTS ARRIVALS EVEN WHEN THE GROUP IS POISONED: a rank that wakes with
    // ZP_ERR_COMM must not return (its caller may free or reuse the buffer it published) while a third rank is still copying from that
    // buffer -- every rank synchronises its own stream first, then waits here until all of them have, or for drain_ms at most (a rank that
    // never comes).  Returns what barrier() would: ZP_OK only when everybody arrived and nobody failed.
    int drain_waiting = 0;
    unsigned drain_gen = 0;
    int drain_ms = 5000;
    int32_t closing_barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned g = drain_gen;
        if (++drain_waiting == world) {
            drain_waiting = 0;
            drain_gen++;
            cv.notify_all();
            return poisoned ? ZP_ERR_COMM : ZP_OK;
        }
        const int wait_ms = poisoned ? (drain_ms < timeout_ms ? drain_ms : timeout_ms) : timeout_ms;
        auto arrived = [&] { return drain_gen != g; };
        bool woke = cv.wait_for(lk, std::chrono::milliseconds(wait_ms), [&] { return arrived() || poisoned; });
        if (woke && !arrived() && poisoned)        // poisoned while waiting: give the ranks still copying drain_ms to arrive
            woke = cv.wait_for(lk, std::chrono::milliseconds(drain_ms < timeout_ms ? drain_ms : timeout_ms), arrived);
        if (!arrived()) {       // somebody never came: the group is dead (and this generation's count is void)
            poisoned = true;
            cv.notify_all();
            return ZP_ERR_COMM;
        }
        return poisoned ? ZP_ERR_COMM : ZP_OK;
    }
};

struct zp_comm {
    zp_ctx *ctx;
    ncclComm_t comm;
    int rank, world;
    zp_comm_group *local = nullptr;    // non-null: in-process group, no RCCL
    bool dead = false;                 // aborted (zp_comm_abort, the watchdog): every later collective is ZP_ERR_COMM
    int timeout_ms = 120000;           // RCCL: > 0 = a collective returns when it is complete on the stream, or ZP_ERR_COMM after this long
    // exchange buffers of zp_merkle_commit_sharded (pack, rows, sub-roots, tree top): grown on demand, kept until zp_comm_destroy.  A commitment
    // per call used to hipMalloc / hipFree two buffers of the matrix's size (8.6 GB each at 2^25 x 32): one call in a dozen then took seconds
    // (round 5: max 4 515 ms against a median of 81) while the driver gave the pages back and mapped them again.
    void *xbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t xbytes[4] = {0, 0, 0, 0};
};

namespace {

__global__ void __launch_bounds__(256) sum_parts_kernel(u64 *__restrict__ dst, const u64 *__restrict__ parts, size_t words, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= words) return;
    u64 a = 0;
    for (int k = 0; k < n; k++) a += parts[(size_t)k * words + i];
    dst[i] = a;
}

int32_t comm_scratch(zp_comm *c, int which, size_t bytes, void **out) {
    if (c->xbytes[which] < bytes) {
        if (c->xbuf[which]) (void)zp_dev_free(c->ctx, c->xbuf[which]);
        c->xbuf[which] = nullptr;
        c->xbytes[which] = 0;
        ZP_TRY(zp_dev_alloc(c->ctx, bytes, &c->xbuf[which]));
        c->xbytes[which] = bytes;
    }
    *out = c->xbuf[which];
    return ZP_OK;
}

const char *const kDeadText = "communicator is dead: a peer rank failed, aborted or did not arrive in time";

// in-process collectives: publish this rank's buffer, meet, copy what the collective says from the peers' buffers, meet again (so that
// nobody reuses a buffer a peer is still reading).  All copies run on the calling rank's stream and are complete on return.  A rank
// whose own step fails still tells the others (poison) before it returns: no rank is left waiting at a barrier.
int32_t local_exchange(zp_comm *c, const void *mine, const std::function<int32_t(const void *const *)> &copy) {
    zp_comm_group *g = c->local;
    int32_t rc = ZP_OK;
    hipError_t e = hipStreamSynchronize(c->ctx->stream);       // what I hand out is complete
    if (e != hipSuccess) {
        c->ctx->err = std::string("hipStreamSynchronize before a collective: ") + hipGetErrorString(e);
        rc = ZP_ERR_HIP;
        g->fail();
    }
    g->ptr[c->rank] = mine;
    const int32_t b1 = g->barrier();
    if (b1 == ZP_OK && rc == ZP_OK) {
        rc = copy(g->ptr);
        e = hipStreamSynchronize(c->ctx->stream);
        if (rc == ZP_OK && e != hipSuccess) {
            c->ctx->err = std::string("hipStreamSynchronize after a collective: ") + hipGetErrorString(e);
            rc = ZP_ERR_HIP;
        }
        if (rc != ZP_OK) g->fail();
    }
    // every rank's own copies are complete here (or were never issued); nobody leaves before every peer has stopped reading.  The opening
    // barrier fails only when its generation never completed (poisoned before the last arrival, or a rank that never came): then no rank of
    // that generation was told to copy, and there is nothing to wait for.
    const int32_t b2 = (b1 == ZP_OK) ? g->closing_barrier() : ZP_ERR_COMM;
    if (rc != ZP_OK) return rc;
    if (b1 != ZP_OK || b2 != ZP_OK) {
        c->ctx->err = kDeadText;
        return ZP_ERR_COMM;
    }
    return ZP_OK;
}

// RCCL: the collective is on the stream.  With a timeout the call returns when the stream has drained -- or aborts the communicator when
// a peer never joins (its kernel would spin for ever and hipStreamSynchronize with it) or RCCL reports an asynchronous error.
int32_t rccl_wait(zp_comm *c) {
    if (c->timeout_ms <= 0) return ZP_OK;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    for (;;) {
        const hipError_t q = hipStreamQuery(c->ctx->stream);
        if (q == hipSuccess) return ZP_OK;
        if (q != hipErrorNotReady) {
            c->ctx->err = std::string("stream failed during a collective: ") + hipGetErrorString(q);
            c->dead = true;
            if (rccl().CommAbort && c->comm) { (void)rccl().CommAbort(c->comm); c->comm = nullptr; }
            return ZP_ERR_HIP;
        }
        ncclResult_t ae = ncclSuccess;
        const bool async_bad = rccl().CommGetAsyncError && (spins & 255) == 255 && rccl().CommGetAsyncError(c->comm, &ae) == ncclSuccess && ae != ncclSuccess &&
                               ae != ncclInProgress;
        const bool late = std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(c->timeout_ms);
        if (async_bad || late) {
            c->ctx->err = async_bad ? std::string("RCCL asynchronous error: ") + rccl().GetErrorString(ae)
                                    : std::string("collective did not complete within the communicator's timeout; communicator aborted");
            c->dead = true;
            if (rccl().CommAbort && c->comm) { (void)rccl().CommAbort(c->comm); c->comm = nullptr; }
            return ZP_ERR_COMM;
        }
        if (++spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

}  // namespace

zp_ctx *zpi_comm_ctx(const zp_comm *comm) { return comm ? comm->ctx : nullptr; }

// this rank cannot go on (rc != ZP_OK, ctx->err already says why): take the communicator down so that no peer waits for it, return rc
int32_t zpi_comm_fail(zp_comm *c, int32_t rc) {
    if (!c || rc == ZP_OK) return rc;
    c->dead = true;
    if (c->local) c->local->fail();
    else if (c->comm && rccl().CommAbort) { (void)rccl().CommAbort(c->comm); c->comm = nullptr; }
    return rc;
}

#define ZP_NCCL(c, call)                                                                       \
    do {                                                                                       \
        const ncclResult_t r_ = (call);                                                        \
        if (r_ != ncclSuccess) {                                                               \
            (c)->ctx->err = std::string(#call) + ": " + rccl().GetErrorString(r_);             \
            return zpi_comm_fail((c), ZP_ERR_HIP);                                             \
        }                                                                                      \
    } while (0)

extern "C" {

int32_t zp_comm_unique_id(uint8_t *out128) {
    if (!out128 || !rccl().ok) return out128 ? ZP_ERR_UNSUPPORTED : ZP_ERR_ARG;
    ncclUniqueId id;
    if (rccl().GetUniqueId(&id) != ncclSuccess) return ZP_ERR_HIP;
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(out128, &id, 128);
    return ZP_OK;
}

int32_t zp_comm_create(zp_ctx *ctx, int32_t rank, int32_t world, const uint8_t *id128, zp_comm **out) {
    if (!ctx || !out) return ZP_ERR_ARG;
    *out = nullptr;
    ZP_BIND(ctx);
    ZP_ARG(ctx, id128 && world >= 1 && rank >= 0 && rank < world, "bad rank / world / id");
    ZP_ARG(ctx, (world & (world - 1)) == 0, "world must be a power of two (row shards are halves of halves of the domain)");
    if (!rccl().ok) { ctx->err = "librccl.so could not be loaded"; return ZP_ERR_UNSUPPORTED; }
    zp_comm *c = new (std::nothrow) zp_comm();
    if (!c) return ZP_ERR_NOMEM;
    c->ctx = ctx; c->rank = rank; c->world = world; c->comm = nullptr;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    const ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        ctx->err = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
        delete c;
        return ZP_ERR_HIP;
    }
    *out = c;
    return ZP_OK;
}

int32_t zp_comm_group_create(int32_t world, zp_comm_group **out) {
    if (!out || world < 1 || world > 64 || (world & (world - 1)) != 0) return ZP_ERR_ARG;
    zp_comm_group *g = new (std::nothrow) zp_comm_group();
    if (!g) return ZP_ERR_NOMEM;
    g->world = world;
    *out = g;
    return ZP_OK;
}

int32_t zp_comm_group_destroy(zp_comm_group *g) {
    delete g;
    return ZP_OK;
}

int32_t zp_comm_create_local(zp_ctx *ctx, int32_t rank, zp_comm_group *group, zp_comm **out) {
    if (!ctx || !out) return ZP_ERR_ARG;
    *out = nullptr;
    ZP_ARG(ctx, group && rank >= 0 && rank < group->world, "bad rank / group");
    zp_comm *c = new (std::nothrow) zp_comm();
    if (!c) return ZP_ERR_NOMEM;
    c->ctx = ctx; c->rank = rank; c->world = group->world; c->comm = nullptr; c->local = group;
    *out = c;
    return ZP_OK;
}

int32_t zp_comm_destroy(zp_comm *c) {
    if (!c) return ZP_OK;
    ZP_BIND(c->ctx);
    if (!c->dead) (void)hipStreamSynchronize(c->ctx->stream);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    for (void *b : c->xbuf)
        if (b) (void)zp_dev_free(c->ctx, b);
    delete c;
    return ZP_OK;
}

// gives the exchange buffers the communicator keeps between commitments back to the device (they grow to the largest matrix committed)
int32_t zp_comm_release_scratch(zp_comm *c) {
    if (!c) return ZP_ERR_ARG;
    ZP_BIND(c->ctx);
    for (int i = 0; i < 4; i++) {
        if (c->xbuf[i]) (void)zp_dev_free(c->ctx, c->xbuf[i]);
        c->xbuf[i] = nullptr;
        c->xbytes[i] = 0;
    }
    return ZP_OK;
}

// What the TRANSPORT says about this communicator, not what its creator was told: out4 = {transport (1 = RCCL, 0 = in-process group),
// ranks the communicator joined (ncclCommCount), this rank inside it (ncclCommUserRank), HIP device it is bound to (ncclCommCuDevice)}.
// A launcher that meant to start 8 ranks and started 8 worlds of one reads 1 here, whatever WORLD_SIZE said.  -1: RCCL does not export the query.
int32_t zp_comm_info(const zp_comm *c, int32_t *out4) {
    if (!c || !out4) return ZP_ERR_ARG;
    if (c->local) {
        out4[0] = 0; out4[1] = c->local->world; out4[2] = c->rank; out4[3] = c->ctx->device;
        return ZP_OK;
    }
    if (c->dead || !c->comm) { c->ctx->err = kDeadText; return ZP_ERR_COMM; }
    int v = -1;
    out4[0] = 1;
    out4[1] = (rccl().CommCount && rccl().CommCount(c->comm, &v) == ncclSuccess) ? v : -1;
    out4[2] = (rccl().CommUserRank && rccl().CommUserRank(c->comm, &v) == ncclSuccess) ? v : -1;
    out4[3] = (rccl().CommCuDevice && rccl().CommCuDevice(c->comm, &v) == ncclSuccess) ? v : -1;
    return ZP_OK;
}

// A rank that fails BETWEEN collectives (out of memory, a bad argument only it sees) calls this before it gives up: in-process peers wake
// from their barrier with ZP_ERR_COMM, RCCL peers run into their timeout (the aborted rank's kernels are gone).  Idempotent.
int32_t zp_comm_abort(zp_comm *c) {
    if (!c) return ZP_ERR_ARG;
    (void)zpi_comm_fail(c, ZP_ERR_COMM);
    return ZP_OK;
}

// how long a collective may wait for its peers (default 120 000 ms).  In-process group: the barrier timeout of the WHOLE group.  RCCL:
// this rank's watchdog; 0 = asynchronous collectives (return after the enqueue, no watchdog: the caller synchronises the stream itself).
int32_t zp_comm_set_timeout_ms(zp_comm *c, int32_t ms) {
    if (!c || ms < 0) return ZP_ERR_ARG;
    if (c->local) {
        if (ms == 0) return ZP_ERR_ARG;
        std::lock_guard<std::mutex> lk(c->local->mu);
        c->local->timeout_ms = ms;
    } else {
        c->timeout_ms = ms;
    }
    return ZP_OK;
}

int32_t zp_comm_rank(const zp_comm *c) { return c ? c->rank : -1; }
int32_t zp_comm_world(const zp_comm *c) { return c ? c->world : -1; }

// recv[h * words_per_peer ..) <- rank h's send[me * words_per_peer ..): one grouped send/recv, all links busy at once
int32_t zp_comm_all_to_all(zp_comm *c, const uint64_t *d_send, uint64_t *d_recv, size_t words_per_peer) {
    if (!c) return ZP_ERR_ARG;
    ZpStage stage_(c->ctx, "comm_all_to_all");
    ZP_ARG(c->ctx, d_send && d_recv && d_send != d_recv, "bad buffers");
    if (words_per_peer == 0) return ZP_OK;
    if (c->dead || (c->local && c->local->dead())) { c->ctx->err = kDeadText; return ZP_ERR_COMM; }
    if (c->local)
        return local_exchange(c, d_send, [&](const void *const *peer) -> int32_t {
            for (int h = 0; h < c->world; h++)
                ZP_HIP(c->ctx, hipMemcpyAsync(d_recv + (size_t)h * words_per_peer, (const u64 *)peer[h] + (size_t)c->rank * words_per_peer,
                                              words_per_peer * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
            return ZP_OK;
        });
    ZP_NCCL(c, rccl().GroupStart());
    ncclResult_t bad = ncclSuccess;       // a failed send / recv must not leave this thread's RCCL group open: GroupEnd runs either way
    const char *what = "";
    for (int h = 0; h < c->world && bad == ncclSuccess; h++) {
        bad = rccl().Send(d_send + (size_t)h * words_per_peer, words_per_peer, ncclUint64, h, c->comm, c->ctx->stream);
        what = "ncclSend";
        if (bad != ncclSuccess) break;
        bad = rccl().Recv(d_recv + (size_t)h * words_per_peer, words_per_peer, ncclUint64, h, c->comm, c->ctx->stream);
        what = "ncclRecv";
    }
    const ncclResult_t ge = rccl().GroupEnd();
    if (bad != ncclSuccess || ge != ncclSuccess) {
        c->ctx->err = std::string(bad != ncclSuccess ? what : "ncclGroupEnd") + ": " + rccl().GetErrorString(bad != ncclSuccess ? bad : ge);
        return zpi_comm_fail(c, ZP_ERR_HIP);
    }
    return rccl_wait(c);
}

int32_t zp_comm_all_gather(zp_comm *c, const uint64_t *d_send, uint64_t *d_recv, size_t words) {
    if (!c) return ZP_ERR_ARG;
    ZpStage stage_(c->ctx, "comm_all_gather");
    ZP_ARG(c->ctx, d_send && d_recv, "bad buffers");
    if (words == 0) return ZP_OK;
    if (c->dead || (c->local && c->local->dead())) { c->ctx->err = kDeadText; return ZP_ERR_COMM; }
    if (c->local)
        return local_exchange(c, d_send, [&](const void *const *peer) -> int32_t {
            for (int h = 0; h < c->world; h++)
                ZP_HIP(c->ctx, hipMemcpyAsync(d_recv + (size_t)h * words, peer[h], words * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
            return ZP_OK;
        });
    ZP_NCCL(c, rccl().AllGather(d_send, d_recv, words, ncclUint64, c->comm, c->ctx->stream));
    return rccl_wait(c);
}

// d_buf <- sum over the ranks of their d_buf (64-bit wrapping sums; the sharded prover adds vectors of which exactly one rank holds a
// non-zero entry per position: the owner of a row answers, the others contribute zeros)
int32_t zp_comm_all_reduce_sum(zp_comm *c, uint64_t *d_buf, size_t words) {
    if (!c) return ZP_ERR_ARG;
    ZpStage stage_(c->ctx, "comm_all_reduce");
    ZP_ARG(c->ctx, d_buf != nullptr, "bad buffer");
    if (words == 0) return ZP_OK;
    if (c->dead || (c->local && c->local->dead())) { c->ctx->err = kDeadText; return ZP_ERR_COMM; }
    if (c->local) {
        void *tmp = nullptr;
        {   // allocate BEFORE the ranks meet; a rank that cannot must say so, or its peers would wait at the barrier
            const int32_t arc = zp_dev_alloc(c->ctx, (size_t)c->world * words * 8, &tmp);
            if (arc != ZP_OK) return zpi_comm_fail(c, arc);
        }
        int32_t rc = local_exchange(c, d_buf, [&](const void *const *peer) -> int32_t {
            for (int h = 0; h < c->world; h++)
                ZP_HIP(c->ctx, hipMemcpyAsync((u64 *)tmp + (size_t)h * words, peer[h], words * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
            return ZP_OK;
        });
        if (rc == ZP_OK) {     // every rank has read every buffer (second barrier of the exchange): now the sums may overwrite them
            hipLaunchKernelGGL(sum_parts_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, c->ctx->stream, (u64 *)d_buf, (const u64 *)tmp, words,
                               c->world);
            if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->ctx->stream) != hipSuccess) rc = zpi_comm_fail(c, ZP_ERR_HIP);
        }
        (void)zp_dev_free(c->ctx, tmp);
        return rc;
    }
    ZP_NCCL(c, rccl().AllReduce(d_buf, d_buf, words, ncclUint64, ncclSum, c->comm, c->ctx->stream));
    return rccl_wait(c);
}

int32_t zp_comm_broadcast(zp_comm *c, uint64_t *d_buf, size_t words, int32_t root) {
    if (!c) return ZP_ERR_ARG;
    ZpStage stage_(c->ctx, "comm_broadcast");
    ZP_ARG(c->ctx, d_buf && root >= 0 && root < c->world, "bad arguments");
    if (words == 0) return ZP_OK;
    if (c->dead || (c->local && c->local->dead())) { c->ctx->err = kDeadText; return ZP_ERR_COMM; }
    if (c->local)
        return local_exchange(c, d_buf, [&](const void *const *peer) -> int32_t {
            if (c->rank != root) ZP_HIP(c->ctx, hipMemcpyAsync(d_buf, peer[root], words * 8, hipMemcpyDeviceToDevice, c->ctx->stream));
            return ZP_OK;
        });
    ZP_NCCL(c, rccl().Broadcast(d_buf, d_buf, words, ncclUint64, root, c->comm, c->ctx->stream));
    return rccl_wait(c);
}

// column shards -> row shards: d_cols u64[Wl][M] (this rank's columns, all rows) -> d_rows u64[G * Wl][M / G] (ALL columns in
// rank order, this rank's rows): zp_pack_blocks, then one all-to-all; the received buffer already is the column-major matrix
// zp_merkle_commit takes.  d_pack: scratch of Wl * M words.
int32_t zp_exchange_columns_to_rows(zp_comm *c, const uint64_t *d_cols, size_t Wl, size_t M, uint64_t *d_pack, uint64_t *d_rows) {
    if (!c) return ZP_ERR_ARG;
    zp_ctx *ctx = c->ctx;
    ZP_ARG(ctx, d_cols && d_pack && d_rows && Wl >= 1 && M % (size_t)c->world == 0 && (M / c->world) % 2 == 0, "bad arguments (M must split into even row shards)");
    if (c->world == 1) return zp_d2d(ctx, d_rows, d_cols, Wl * M * 8);
    const int32_t prc = zp_pack_blocks(ctx, d_cols, d_pack, Wl, M, c->world);
    if (prc != ZP_OK) return zpi_comm_fail(c, prc);        // the peers are on their way into the all-to-all
    return zp_comm_all_to_all(c, d_pack, d_rows, Wl * (M / c->world));
}

// One Merkle commitment over the ranks: every rank hands in its Wl columns of the M-row matrix and gets the GLOBAL root (the
// same root a single GPU computes over all G * Wl columns).  d_tree_local receives this rank's subtree over its M / G rows
// ((2 M / G - 1) * 4 words, what zp_merkle_open_batch answers openings of local rows from); the G sub-roots are all-gathered
// and the top log2 G levels hashed redundantly on every rank.
int32_t zp_merkle_commit_sharded(zp_comm *c, const uint64_t *d_cols, size_t M, int32_t Wl, uint64_t *d_tree_local, uint64_t *h_root4) {
    if (!c) return ZP_ERR_ARG;
    zp_ctx *ctx = c->ctx;
    ZP_ARG(ctx, d_cols && d_tree_local && h_root4 && Wl >= 1 && M >= (size_t)2 * c->world && (M & (M - 1)) == 0, "bad arguments");
    const size_t G = (size_t)c->world, Ml = M / G;
    void *pack = nullptr, *rows = nullptr, *sub = nullptr;
    int32_t rc = comm_scratch(c, 0, G > 1 ? (size_t)Wl * M * 8 : 8, &pack);      // a world of one copies, it does not pack
    if (rc == ZP_OK) rc = comm_scratch(c, 1, (size_t)Wl * M * 8, &rows);
    if (rc == ZP_OK) rc = comm_scratch(c, 2, G * 12 * 8, &sub);
    if (rc == ZP_OK) rc = zp_exchange_columns_to_rows(c, d_cols, (size_t)Wl, M, (uint64_t *)pack, (uint64_t *)rows);
    if (rc == ZP_OK) rc = zp_merkle_commit(ctx, (const uint64_t *)rows, Ml, (int32_t)(G * Wl), d_tree_local);
    std::vector<u64> lvl(4);
    void *top = nullptr;                                     // the top log2(G) levels: the tree over the G sub-roots, on the device in one call
    if (rc == ZP_OK) rc = zp_comm_all_gather(c, d_tree_local + (2 * Ml - 2) * 4, (uint64_t *)sub, 4);
    if (rc == ZP_OK) rc = comm_scratch(c, 3, (2 * G - 1) * 32, &top);
    if (rc == ZP_OK) rc = zp_merkle_commit_rows(ctx, (const uint64_t *)sub, G, 4, (uint64_t *)top);       // leaves of 4 values are their own digests
    if (rc == ZP_OK) rc = zp_d2h(ctx, lvl.data(), (const uint64_t *)top + (2 * G - 2) * 4, 32);
    if (rc == ZP_OK) memcpy(h_root4, lvl.data(), 32);
    else (void)zpi_comm_fail(c, rc);                       // whatever failed here, no peer may wait for this rank
    return rc;
}

// this rank's rows u64[Rl][C] of an (G Rl) x C matrix -> this rank's rows u64[C / G][G Rl] of its transpose: pack + ONE all-to-all
// (zp_exchange_columns_to_rows) + the local transpose.  d_x is clobbered (it receives the exchanged blocks), d_z is the pack buffer.
static int32_t sharded_transpose(zp_comm *c, uint64_t *d_x, uint64_t *d_y, uint64_t *d_z, size_t Rl, size_t C) {
    zp_ctx *ctx = c->ctx;
    const size_t G = (size_t)c->world;
    if (G == 1) return zp_transpose(ctx, d_x, d_y, Rl, C);
    const int32_t prc = zp_pack_blocks(ctx, d_x, d_z, Rl, C, c->world);
    if (prc != ZP_OK) return zpi_comm_fail(c, prc);
    ZP_TRY(zp_comm_all_to_all(c, d_z, d_x, Rl * (C / G)));            // block h of d_x = rank h's rows, my columns: [G Rl][C / G]
    return zp_transpose(ctx, d_x, d_y, G * Rl, C / G);
}

// Four-step NTT of ONE column of N = 2^logn elements split over the ranks (BASELINE.json configs[3] / SURVEY.md 8e: "RCCL all-to-all
// over xGMI for the four-step NTT transpose").  N = N1 N2 (N1 = 2^(logn / 2)), input index i = i1 N2 + i2, output k = k1 + N1 k2:
//     X[k1 + N1 k2] = sum_i2 w_N2^(i2 k2) w_N^(i2 k1) sum_i1 x[i1 N2 + i2] w_N1^(i1 k1)
// Rank g holds the contiguous block [g N / G, (g + 1) N / G) of the column in d_data = rows i1 of the N1 x N2 matrix.  Steps:
// transpose (all-to-all) -> N1-point transforms of the local N2 / G rows (zp_ntt) -> twiddle w_N^(i2 k1) (zp_twiddle_rows) ->
// transpose (all-to-all) -> N2-point transforms of the local N1 / G rows -> with natural_output a third transpose, so that d_data
// ends as this rank's contiguous block of the transform; without it d_data holds rows k1 in [g N1 / G, ..) of Y[k1][k2] =
// X[k1 + N1 k2] (what a consumer that works on rows wants: one all-to-all less).  inverse != 0: the inverse transform, scaled by
// 1 / N.  d_tmp: scratch of 2 N / G words.  Same result, bit for bit, as zp_ntt / zp_intt on the whole column on one GPU.
static int32_t ntt_sharded_impl(zp_comm *c, uint64_t *d_data, uint64_t *d_tmp, int32_t logn, int32_t inverse, int32_t natural_output);
int32_t zp_ntt_sharded(zp_comm *c, uint64_t *d_data, uint64_t *d_tmp, int32_t logn, int32_t inverse, int32_t natural_output) {
    if (!c) return ZP_ERR_ARG;
    return zpi_comm_fail(c, ntt_sharded_impl(c, d_data, d_tmp, logn, inverse, natural_output));   // an error on this rank frees the peers
}
static int32_t ntt_sharded_impl(zp_comm *c, uint64_t *d_data, uint64_t *d_tmp, int32_t logn, int32_t inverse, int32_t natural_output) {
    zp_ctx *ctx = c->ctx;
    const size_t G = (size_t)c->world;
    const int l1 = logn / 2, l2 = logn - l1;
    ZP_ARG(ctx, d_data && d_tmp && logn >= 2 && logn <= 40 && (G & (G - 1)) == 0 && ((size_t)1 << l1) >= 2 * G, "bad arguments (2^(logn/2) must be at least twice the world size)");
    const size_t N1 = (size_t)1 << l1, N2 = (size_t)1 << l2, Nl = (N1 / G) * N2;
    uint64_t *X = d_data, *Y = d_tmp, *Z = d_tmp + Nl;
    ZP_TRY(sharded_transpose(c, X, Y, Z, N1 / G, N2));                         // Y [N2 / G][N1]: row i2, entries over i1
    ZP_TRY((inverse ? zp_intt : zp_ntt)(ctx, Y, Y, l1, (int32_t)(N2 / G)));    // over i1 -> k1
    ZP_TRY(zp_twiddle_rows(ctx, Y, l1, (int32_t)(N2 / G), (uint64_t)c->rank * (N2 / G), logn, inverse));
    ZP_TRY(sharded_transpose(c, Y, X, Z, N2 / G, N1));                         // X [N1 / G][N2]: row k1, entries over i2
    ZP_TRY((inverse ? zp_intt : zp_ntt)(ctx, X, X, l2, (int32_t)(N1 / G)));    // over i2 -> k2: Y[k1][k2]
    if (!natural_output) return ZP_OK;
    ZP_TRY(sharded_transpose(c, X, Y, Z, N1 / G, N2));                         // Y [N2 / G][N1]: k = k1 + N1 k2, my rows k2
    return zp_d2d(ctx, X, Y, Nl * 8);
}

}  // extern "C"
