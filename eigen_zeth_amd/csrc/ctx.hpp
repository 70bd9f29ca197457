// Internal context shared by the translation units of libzethprover.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/zeth_prover.h"
#include "gl.hpp"

#define ZP_ROOT32_DEFAULT 1753635133440165772ULL /* 7^((p-1)/2^32), SURVEY.md 8a-N1 */
#define ZP_SHIFT_DEFAULT 49ULL

struct NttPass {
    int L, A1, A2, A3, logT;
    int logPprev;  // log2 of the product of the radices of the previous passes
};

struct NttPlan {
    int logn = 0;
    bool inverse = false;
    int npass = 0;
    NttPass pass[6];
    int lb = 0;               // twiddle split: w^e = twl[e & (2^lb-1)] * twh[e >> lb]
    u64 *d_twl = nullptr;     // 2^lb entries
    u64 *d_twh = nullptr;     // 2^(logn-lb) entries
    u64 *d_tws = nullptr;     // w_4096^e (direction matched), 4096 entries
    u64 *d_tw1 = nullptr;     // first (transposing) pass: w^(u k) at [u * R + k], all 2^logn of them (null: per-lane chains)
    bool tw1_unavailable = false;   // its allocation failed once: the per-lane chain kernel serves this plan
    u64 w16[8];               // w_16^i (direction matched)
    u64 ninv = 1;
    int j0inv = 0;            // w_16 = (2^12)^j0 ; j0inv = j0^-1 mod 16 (0: not on the power-of-two path)
};

struct CosetTable {  // shift^i * pre, i < 2^logn, two-level
    int logn = 0, lb = 0;
    u64 shift = 0, pre = 0;
    u64 *d_lo = nullptr, *d_hi = nullptr;
};

struct ZpG16Cache;
struct zp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // where every launch of this ctx goes
    hipStream_t own_stream = nullptr;  // created by zp_create (non-blocking); replaced, not destroyed, by zp_set_stream
    std::string err;
    u64 root32 = ZP_ROOT32_DEFAULT;
    u64 coset_shift = ZP_SHIFT_DEFAULT;
    // Poseidon tables (device copies)
    u64 h_rc[360];
    u64 h_mds[144];
    u64 *d_rc = nullptr;
    u32 *d_mds = nullptr;
    bool poseidon_dirty = true;
    bool mds_is_default = false;  // compile-time literal fast path for the documented default matrix
    // plans
    std::map<int, NttPlan> plans;  // key = logn*2 + inverse
    std::vector<CosetTable> cosets;
    // scratch (two ping-pong buffers, grown on demand)
    u64 *scratch[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // 0, 1: NTT ping-pong; 2: LDE coefficients; 3: small uploads; 4: fixed-column periods; 5: zp_ood_eval's weight vector
    size_t scratch_elems[6] = {0, 0, 0, 0, 0, 0};
    // zp_ood_eval: the barycentric weights last made (scratch 5) are for this domain size and point
    bool ood_valid = false;
    int ood_logn = -1;
    u64 ood_y[3] = {0, 0, 0}, ood_root32 = 0;
    // pinned host staging for small transfers (pageable async copies lock/unlock host pages on every call)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    zp_ctx *msm_helpers[4] = {nullptr, nullptr, nullptr, nullptr};   // zp_groth16_prove: ctxs (streams, MSM arenas) of the MSMs that run beside the first one
    ZpG16Cache *g16_cache = nullptr;   // zp_r1cs_eval_device: the circuit last used, in HBM (csrc/r1cs.hip; freed by zpi_g16_cache_free)
    int num_cu = 256;
    // experiment knobs (zp_set_tuning): not part of the stable surface
    void *msm_arena = nullptr;    // scratch of zp_msm_bn254*: grows to the largest run, freed by zp_destroy
    size_t msm_arena_bytes = 0;
    int tune_logt12 = 2;          // radix-4096 passes: 2 = tiles of 4 columns (128 KiB, one 1024-thread workgroup per CU), 1 = tiles of 2 columns (64 KiB, two 512-thread workgroups per CU; A/B: profiles/r6_ntt_two_pass_2wg_ab.txt)
    int tune_logt = 4, tune_tpw = 2, tune_logt9 = 4;   // (radix-512 passes: 16-wide tiles of 64 KiB, two workgroups per CU -- 5 to 19 % faster than the 32-wide 128-KiB tile at 2^17 .. 2^27, profiles/r4_logt9_sweep.txt)
    // tiles per workgroup: 2 measured best with the gl_asm.hpp arithmetic (profiles/r2_ntt_sweep.txt)
    int tune_ntt_tw1 = 26;        // largest log size whose first pass multiplies by a full precomputed table (0: always the per-lane chain)
    int tune_ntt_limb = 0;        // 1: register butterflies on the four-limb form of gl_limb.hpp (bit-identical, 17-35 % fewer VALU cycles, but 168 VGPRs and 48 KiB of LDS: 3 workgroups per CU instead of 4 -- measured 4 % slower, profiles/r4_ntt_limb_ab.txt)
    int tune_lde_seam = 1;        // blow-up 2 with radix-256 passes on both sides of the seam: the inverse transform's last pass and the forward one's first in ONE kernel (0: two launches through the coefficient buffer)
    int tune_seam_tpw = 2;        // inverse tiles per workgroup of the seam kernel
    int tune_copy_grid = 0, tune_copy_nt = 1, tune_copy_block = 0, tune_copy_unroll = 0;   // zp_hbm_copy_probe: workgroups (0 = 256: one per CU), policy (1 = non-temporal), lanes per workgroup (0 = 256), 16-byte loads in flight per lane (0 = 4)
    int tune_lde_seam_plans = 1;  // the fused extension may use plans made for it (an inverse plan ending / a forward plan starting in a radix-256 pass) where the default plans do not meet in one (0: default plans only)
    int tune_g16_parallel = 1;    // zp_groth16_prove: the five MSMs of a proof on five streams at once (0: one after the other)
    int tune_ntt_order = 0;       // plan digit order: 0 auto (a radix-512 digit goes last), 1 larger radices first, 2 larger radices last
    int tune_ntt_maxl = 0;        // 0 = 9: largest log2 radix of one NTT pass (10: 1024-thread workgroups, two-pass plans up to 2^20)
    int tune_ntt_small_wave = 0;  // transforms of <= 4096 points: their last six stages inside a wave (DPP / ds_swizzle / ds_bpermute lane exchanges) instead of LDS + a barrier per stage.  0: where it measured faster (<= 64 points), 1: always, 2: never (A/B: profiles/r5_dpp_ab.txt)
    int tune_fri_fold_lanes = 0;  // the FRI fold by 16 with a coset spread over the 16 lanes of a DPP row.  0: where it measured faster (<= 2^16 inputs), 1: always, 2: never (A/B: profiles/r5_dpp_ab.txt)
    int tune_synth_rowwise = 0;   // the synthetic witness's expansion: 1: one 8-byte store per lane and row (round 4), 2: rows staged through LDS and written as column runs, 0: the faster of the two by size (A/B: profiles/r5_synth_fill_ab.txt)
    int tune_p254_block = 0;     // 2: the lane-per-permutation Poseidon-BN254 kernel walks the partial rounds one by one instead of in blocks of four (A/B)
    int tune_p254_scaled = 0;     // 2: the cooperative Poseidon-BN254 kernels walk the unscaled sparse partial rounds (five dependent products per round instead of three; A/B: profiles/r5_p254_scaled_ab.txt)
    int tune_merkle_top_wave = 0; // 1: the subtree kernel of the small tree levels exchanges a node's state by wave shuffles (no LDS, one barrier per level) instead of LDS + two barriers per round
    int tune_merkle_coop_log = 0; // 0 = 15: tree levels with <= 2^15 nodes go to the 12-lanes-per-node subtree kernel
    int tune_ntt_chunk_log = 0;   // 0 = 28: columns per launch such that a ping-pong scratch buffer is <= 2 GiB
    int tune_p254_bulk_log = 0;   // 0 = 14: Poseidon-BN254 t = 17 launches of >= 2^14 permutations use the lane-per-permutation kernel (31 = never)
    int tune_msm_c = 0;           // 0 = window width chosen from n
    int tune_msm_chunk_log = 0;   // 0 = default (2^24 points per Pippenger run)
    // zp_stark_prove: device buffers kept between proofs (chunk after chunk has the same shapes; hipMalloc / hipFree of ~20 buffers
    // cost milliseconds per proof) and the LDE of the two boundary selectors per (logn, logb, shift, root)
    std::multimap<size_t, void *> prove_pool;
    size_t prove_pool_bytes = 0;
    std::map<std::string, u64 *> prove_fixed;
    size_t prove_fixed_bytes = 0;
    std::map<std::string, void *> air_kernels;   // 64-hex-digit program digest -> generated constraint kernel (AIR plug-in ABI): zp_stark_set_air_kernel
    struct DigestEntry { std::vector<uint64_t> words; uint8_t dg[32]; };
    std::vector<DigestEntry> digest_cache;   // SHA-256 of the large constraint programs seen last (csrc/prove.hip: program_digest)
    std::vector<u64> last_openings;          // BN128-hash mode: roots, query indices, opened values and paths of the last proof, binary (zp_stark_openings)
    // per-launch event profiling (zp_set_profiling)
    bool profiling = false;
    struct PassEv { hipEvent_t a, b; int radix_log; };
    std::vector<PassEv> pass_events;
    struct StageEv { hipEvent_t a, b; const char *name; };
    std::vector<StageEv> stage_events;
};

// brackets one C-ABI compute call with HIP events on the ctx stream when profiling is on (zp_stage_timings)
struct ZpStage {
    zp_ctx *ctx;
    zp_ctx::StageEv ev;
    bool on;
    ZpStage(zp_ctx *c, const char *name) : ctx(c), on(c && c->profiling) {
        if (c) (void)hipSetDevice(c->device);   // a host thread may alternate between ctxs of different GPUs
        if (on) {
            ev.name = name;
            on = hipEventCreate(&ev.a) == hipSuccess && hipEventCreate(&ev.b) == hipSuccess &&
                 hipEventRecord(ev.a, ctx->stream) == hipSuccess;
        }
    }
    ~ZpStage() {
        if (on && hipEventRecord(ev.b, ctx->stream) == hipSuccess) ctx->stage_events.push_back(ev);
    }
};

// first statement of every entry point that touches the device without a ZpStage
#define ZP_BIND(ctx) do { if (ctx) (void)hipSetDevice((ctx)->device); } while (0)

#define ZP_HIP(ctx, call)                                                                    \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return e_ == hipErrorOutOfMemory ? ZP_ERR_NOMEM : ZP_ERR_HIP;                    \
        }                                                                                    \
    } while (0)

#define ZP_ARG(ctx, cond, msg)                                                               \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            (ctx)->err = std::string("bad argument: ") + (msg);                              \
            return ZP_ERR_ARG;                                                               \
        }                                                                                    \
    } while (0)

#define ZP_TRY(expr)                                                                         \
    do {                                                                                     \
        int32_t rc_ = (expr);                                                                \
        if (rc_ != ZP_OK) return rc_;                                                        \
    } while (0)

// internal helpers implemented across the .hip files
int32_t zpi_scratch(zp_ctx *ctx, int which, size_t elems, u64 **out);
int32_t zpi_pinned(zp_ctx *ctx, size_t bytes, void **out);
// small copies through the pinned staging buffer (synchronous on the ctx stream)
int32_t zpi_d2h_small(zp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int32_t zpi_h2d_small(zp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
#define ZP_SMALL_COPY (4u << 20)
int32_t zpi_get_plan(zp_ctx *ctx, int logn, bool inverse, NttPlan **out);
int32_t zpi_get_plan_role(zp_ctx *ctx, int logn, bool inverse, int role, NttPlan **out);   // role 1 / 2: ends / starts with a radix-256 pass (csrc/ntt.hip)
int32_t zpi_lde_plan_json(zp_ctx *ctx, int logn, int want_coef, std::string *out);           // which path zp_lde takes at this size
int32_t zpi_get_coset(zp_ctx *ctx, int logn, u64 shift, u64 pre, CosetTable **out);
int32_t zpi_poseidon_sync_tables(zp_ctx *ctx);
int32_t zpi_poseidon_chains(zp_ctx *ctx, const u64 *d_blocks, const unsigned char *d_absorb, const unsigned int *d_first, int nchains, u64 *d_out);
int32_t zpi_poseidon_openings_walk(zp_ctx *ctx, const u64 *d_op, const u64 *d_vals, u64 mw, const u64 *d_index, const u64 *d_sib, const u64 *d_sib_off,
                                   size_t count, u64 *d_inputs, u64 *d_digests);
// run the transform on W columns; in/out column strides are 2^logn (or in_valid for zero-padded input)
struct NttRunOpts {
    const CosetTable *post_scale = nullptr;  // multiply output i by table(i) (last pass)
    int in_valid_log = -1;                   // >=0: input columns have 2^in_valid_log elements, rest is zero
};
// the sparse periodic fixed columns (2..n_fixed-1) of a constraint program: table behind the stage-2 table,
// per column [lp | n_entries << 8] then n_entries x (pos | is_pub << 63, value or public index)
struct ZpFixedCol { int lp; size_t first_entry_word, n_entries; bool has_pub; };
// validates the whole-blob length and the table; fills `cols` (empty for n_fixed == 2).  false: malformed
bool zpi_program_fixed_table(const uint64_t *h_program, size_t program_words, std::vector<ZpFixedCol> *cols);
// zp_fixed_columns; only_pub: refresh just the columns that hold public inputs inside a buffer whose other columns are already built
int32_t zpi_fixed_columns_build(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub, int32_t logn, int32_t logb,
                                uint64_t shift, uint64_t *d_out, size_t out_words, bool only_pub);
// device buffers from / back to the per-ctx pool of zp_stark_prove (csrc/prove.hip): everything runs on the ctx stream, so reuse is ordered
int32_t zpi_pool_alloc(zp_ctx *ctx, size_t bytes, void **out);
void zpi_pool_release(zp_ctx *ctx, void *p, size_t bytes);
void zpi_sha256(const uint8_t *data, size_t len, uint8_t *out32);
void zpi_g16_cache_free(zp_ctx *ctx);
int32_t zpi_r1cs_poseidon17(zp_ctx *ctx, const u64 *d_inst, size_t i0, size_t count, u64 *d_w, unsigned char *d_set, u64 *d_a, u64 *d_b, u64 *d_c,
                            unsigned long long *d_flags, int *rp_out);
int32_t zpi_merkle16_levels_bn254(zp_ctx *ctx, u64 *d_tree, size_t n);   // the levels above n digests at the head of a 16-ary tree buffer (csrc/poseidon_bn254.hip)
struct zp_comm;
zp_ctx *zpi_comm_ctx(const zp_comm *comm);      // the ctx a communicator was created on (csrc/comm.hip)
int32_t zpi_comm_fail(zp_comm *comm, int32_t rc);   // rc != ZP_OK: kill the communicator (no peer waits for this rank); returns rc
int32_t zpi_twiddle_rows(zp_ctx *ctx, u64 *d_rows, int logn_row, int W, u64 row0, int logn_total, bool inverse);
int32_t zpi_lde(zp_ctx *ctx, const u64 *d_in, u64 *d_out, u64 *d_coef, int logn, int logb, int W, u64 shift);
int32_t zpi_ntt_run(zp_ctx *ctx, const u64 *d_in, u64 *d_out, int logn, int W, bool inverse,
                    const NttRunOpts &opts);
