/*
 * zeth_prover.h -- C ABI of libzethprover.so, the MI355X-native arithmetic back end of the
 * prover.v1.ProverService batch prover that eigen-zeth calls.
 *
 * What this replaces in the reference: nothing *inside* /root/reference -- eigen-zeth holds only
 * the gRPC client (src/prover/provider.rs:603-705 connects, :292-310/:358-377/:422-433/:472-483
 * send the four requests of proto/prover/v1/prover.proto:13-36).  The arithmetic that answers
 * GenChunkProof (prover.proto:56-66,93-111) lives in the external prover service (SURVEY.md par.0.2,
 * row 16 of par.2).  A Rust host (as BASELINE.json's north_star asks) would bind exactly these
 * entry points with `extern "C"` -- see INTEGRATION.md for the stub; in this repo the host side is
 * Python/ctypes (eigen_zeth_amd/native.py) because no Rust toolchain exists in the image.
 *
 * Conventions
 *  - every function returns int32_t: 0 = ZP_OK, <0 = error; zp_last_error(ctx) gives the text.
 *  - no exceptions / C++ types cross the boundary; pointers + sizes only.
 *  - field elements are canonical uint64_t (< p = 2^64 - 2^32 + 1).
 *  - matrices are COLUMN-MAJOR  u64[W][N]  (column c at base + c*N), N a power of two.
 *  - F_{p^3} vectors are plane-major u64[3][n]  (x^3 - x - 1).
 *  - pointers named d_* are DEVICE pointers (hipMalloc / torch.cuda storage); h_* are host.
 *  - a ctx is bound to one HIP device and one stream: zp_create makes a non-blocking stream of its own, so
 *    separate ctxs overlap (one per GPU, or prover + witness upload on one GPU); zp_set_stream replaces it with
 *    the caller's stream (NULL = the legacy default stream), zp_get_stream hands it to code that launches beside
 *    the library (the AIR plug-in kernels).  A ctx is single-threaded; buffers written through one ctx may be
 *    read through another after zp_sync / a synchronous copy on the writer.
 *  - all compute entry points are asynchronous on the ctx stream; zp_sync() waits.
 */
#ifndef ZETH_PROVER_H
#define ZETH_PROVER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zp_ctx zp_ctx;

enum {
    ZP_OK = 0,
    ZP_ERR_ARG = -1,     /* bad argument (size not a power of two, null pointer, ...) */
    ZP_ERR_HIP = -2,     /* HIP runtime error (text in zp_last_error) */
    ZP_ERR_NOMEM = -3,   /* device allocation failed */
    ZP_ERR_UNSUPPORTED = -4,
    ZP_ERR_INTERNAL = -5, /* an internal failure that is not the caller's (never an exception: nothing unwinds across this ABI) */
    ZP_ERR_COMM = -6      /* the communicator is dead: a peer rank failed, aborted or did not reach a collective within the timeout */
};

/* kinds for zp_set_constants */
enum {
    ZP_CONST_ROOT32 = 1,        /* blob[1]: primitive 2^32-th root of unity (default 7^((p-1)/2^32)) */
    ZP_CONST_POSEIDON_RC = 2,   /* blob[360]: round constants, rc[r*12+i] */
    ZP_CONST_POSEIDON_MDS = 3,  /* blob[144]: row-major 12x12, out[r] = sum_j m[r*12+j]*in[j] (entries < 2^28) */
    ZP_CONST_COSET_SHIFT = 4    /* blob[1]: default LDE coset shift (49) */
};

/* ---- lifetime ------------------------------------------------------------------------------ */
int32_t zp_create(zp_ctx **out, int32_t device);
void zp_destroy(zp_ctx *ctx);
const char *zp_last_error(zp_ctx *ctx);
const char *zp_version(void);
int32_t zp_device_count(void);      /* visible HIP devices (0 without a GPU) */
int32_t zp_set_stream(zp_ctx *ctx, void *hip_stream);
int32_t zp_get_stream(zp_ctx *ctx, void **hip_stream);
int32_t zp_sync(zp_ctx *ctx);
int32_t zp_set_constants(zp_ctx *ctx, int32_t kind, const uint64_t *blob, size_t n);
int32_t zp_get_constants(zp_ctx *ctx, int32_t kind, uint64_t *blob, size_t n);

/* ---- device memory (for hosts without their own allocator) ---------------------------------- */
int32_t zp_dev_alloc(zp_ctx *ctx, size_t bytes, void **d_ptr);
int32_t zp_dev_free(zp_ctx *ctx, void *d_ptr);
/* page-locked, device-visible host memory for witnesses: zp_h2d / zp_d2h from it are plain DMA at PCIe rate; a copy
 * from pageable memory is staged by the runtime under a lock that stalls every other thread's HIP calls meanwhile */
int32_t zp_host_alloc(zp_ctx *ctx, size_t bytes, void **h_ptr);
int32_t zp_host_free(zp_ctx *ctx, void *h_ptr);
int32_t zp_h2d(zp_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int32_t zp_d2h(zp_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int32_t zp_d2d(zp_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
int32_t zp_dev_zero(zp_ctx *ctx, void *d_dst, size_t bytes);   /* zero-fill on the ctx stream */

/* ---- N1: Goldilocks NTT / iNTT  (natural order in, natural order out) ------------------------
 * d_in, d_out: u64[W][2^logn].  d_in == d_out is allowed.  d_in is preserved when d_in != d_out. */
int32_t zp_ntt(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t W);
int32_t zp_intt(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t W);

/* four-step NTT of ONE column of 2^logn_total elements that is split over GPUs ("single huge
 * column"): between the two local transforms the N2/G x N1 block a GPU holds is multiplied by w_N^(i2*k1):
 *   d_rows[r][k] *= w_N^((row0 + r) * k),  r < W rows of 2^logn_row elements, N = 2^logn_total
 * (inverse != 0: the inverse root).  The transforms themselves are zp_ntt / zp_intt on the rows; the exchange
 * is the host's (eigen_zeth_amd/multigpu.py: one all-to-all per transpose).                               */
int32_t zp_twiddle_rows(zp_ctx *ctx, uint64_t *d_rows, int32_t logn_row, int32_t W, uint64_t row0, int32_t logn_total,
                        int32_t inverse);

/* layout kernels of the multi-GPU paths (device buffers, out of place):
 * zp_pack_blocks: u64[rows][row_len] -> u64[parts][rows][row_len/parts] (block h = columns [h*len/parts, (h+1)*len/parts) of
 *   every row): the send buffer of the column-shard -> row-shard all-to-all, message h contiguous;
 * zp_transpose:   u64[rows][cols] -> u64[cols][rows]: the local part of a distributed transpose (four-step NTT).        */
int32_t zp_pack_blocks(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t rows, size_t row_len, int32_t parts);
int32_t zp_transpose(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, size_t rows, size_t cols);

/* ---- N2: low-degree extension ----------------------------------------------------------------
 * d_in u64[W][2^logn] evaluations on <w_N>; d_out u64[W][2^(logn+logb)] evaluations on
 * shift*<w_bN>, natural order.  If d_coef != NULL it receives u64[W][2^logn], the coefficients of the
 * interpolant composed with the coset shift: d_coef[c][i] = c_i * shift^i (ascending), i.e. p_c(z) = sum_i d_coef[c][i]
 * (z/shift)^i -- zp_poly_eval_ext at z/shift gives p_c(z); the transform already has this vector in hand, the plain c_i
 * would cost one more pass over the data.  shift == 0 selects the ctx default (ZP_CONST_COSET_SHIFT).              */
int32_t zp_lde(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, uint64_t *d_coef, int32_t logn,
               int32_t logb, int32_t W, uint64_t shift);

/* ---- N3: Poseidon-12 and Merkle commitment -------------------------------------------------- */
/* d_states: u64[count][12], permuted in place */
int32_t zp_poseidon_perm(zp_ctx *ctx, uint64_t *d_states, size_t count);
/* proof-of-work grinding in front of a STARK's query phase: *h_nonce = the smallest n with
 * Poseidon(h_seed4[0..3] || n || 0^7)[0] >> (64 - bits) == 0   (bits in 0..40; 0 returns 0).     */
/* Round-by-round trace of `count` permutations -- the witness columns of a Poseidon AIR (the STARK-verifier AIR of
 * GenAggregatedProof / GenFinalProof, prover.proto:115-148): d_inputs u64[count][12]; permutation k fills rows 32k .. 32k+31 of
 * 12 state columns (d_states + e * stride: the state BEFORE round r on row r < 30, the output on rows 30 and 31) and of 12
 * cube columns (d_cubes + e * stride: (state + round constant)^3, state^3 on rows 30, 31).  stride >= 32 * count, in elements. */
int32_t zp_poseidon_trace(zp_ctx *ctx, const uint64_t *d_inputs, size_t count, uint64_t *d_states, uint64_t *d_cubes, size_t stride);
/* The Fiat-Shamir sponge (rate 8, capacity 4) in one launch: for each of the nblocks blocks of 8 the rate h_state[0..8) is
 * overwritten with the block and the state permuted (nblocks = 0: one permutation); then `extra` more permutations.
 * h_state (12 words) is updated; h_rates receives (1 + extra) * 8 words: the rate after the absorption and after each
 * extra permutation.  One host round trip per transcript step instead of one per permutation.                          */
int32_t zp_poseidon_sponge(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates);
/* the same step, also returning the capacity (state[8..12)) after EVERY permutation in h_caps ((max(nblocks, 1) + extra) * 4 words):
 * with the blocks and the rates that is the input state of every permutation -- the witness of the transcript blocks of the
 * verifier AIR (eigen_zeth_amd/stark/verifier_air.py; GenAggregatedProof, prover.proto:115-126). */
int32_t zp_poseidon_sponge_caps(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates,
                                uint64_t *h_caps);
int32_t zp_pow_grind(zp_ctx *ctx, const uint64_t *h_seed4, int32_t bits, uint64_t *h_nonce);
/* leaf i = linear hash (sponge, rate 8, capacity 4; rows of <= 4 elements are identity-padded) of
 * row i across the W columns of d_cols u64[W][M]; d_tree receives (2M-1)*4 u64: M leaves, then
 * M/2 nodes ... the root is the last 4 elements.                                               */
int32_t zp_merkle_commit(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, uint64_t *d_tree);
/* same, leaves are M contiguous rows of `len` elements (row-major), e.g. FRI layer cosets */
int32_t zp_merkle_commit_rows(zp_ctx *ctx, const uint64_t *d_rows, size_t M, size_t len, uint64_t *d_tree);
/* authentication path (host tree or device tree): depth*4 u64, bottom-up siblings */
int32_t zp_merkle_open(zp_ctx *ctx, const uint64_t *d_tree, size_t M, size_t idx, uint64_t *h_path);

/* ---- BN128-hash mode (the last recursive STARK before the Groth16 wrap): Poseidon over the BN254 scalar field F_r,
 * x^5, 8 full + rp partial rounds, widths t = 3 and t = 17; 16-ary Merkle tree over Goldilocks columns, 56 values to a sponge
 * block of 16 field elements (three per element in bits 0..191, 32-bit halves of values 48..55 in bits 192..223).  Field elements cross the ABI as 4 little-endian u64 words, standard form, < r.
 * zp_set_poseidon_bn254 installs the tables of one width (h_rc: (8 + rp) * t elements round-major, h_mds: t * t row-major);
 * eigen_zeth_amd/poseidon_constants.py:bn254_poseidon_params derives them (Grain LFSR; the t = 3 set reproduces the
 * published vector poseidon([1, 2]) = 0x115cc0f5e7d690413df64c6b9662e9cf2a3617f2743245519e19607a4417189a).
 * zp_poseidon_bn254_perm: d_states u64[count][t][4] permuted in place.
 * zp_merkle16_commit_bn254: leaf i = sponge (16 elements per permutation, capacity = previous digest, first capacity 0)
 *   over row i of d_cols u64[W][M] in blocks of 56 values (a + b 2^64 + c 2^128 + half 2^192); node = digest of [0, 16 children] (missing = 0);
 *   d_tree u64[zp_merkle16_nodes(M)][4]: leaves, then each level, root last.
 * zp_merkle16_open_bn254: h_path u64[levels][16][4] = per level the 16 digests of the group on the path (bottom-up).   */
int32_t zp_set_poseidon_bn254(zp_ctx *ctx, int32_t t, int32_t rp, const uint64_t *h_rc, const uint64_t *h_mds);
int32_t zp_poseidon_bn254_perm(zp_ctx *ctx, uint64_t *d_states, size_t count, int32_t t);
/* the transcript sponge of the BN128 mode (width 17: element 0 capacity, 1..16 rate): absorb nblocks blocks of 16 elements, then
 * `extra` more permutations; h_state 17 elements in/out, h_rates (1 + extra) * 16 elements */
int32_t zp_poseidon_bn254_sponge(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra, uint64_t *h_rates);
/* the same, also handing out the capacity element (element 0) after EVERY permutation, in order: h_caps u64[max(nblocks, 1) + extra][4]
 * (may be NULL).  The wrap circuit re-hashes the final STARK's transcript with all its gadgets side by side: each gadget's capacity input is
 * a caller-set wire, tied to its predecessor's output by a constraint (service/wrap_circuit.py).                                        */
int32_t zp_poseidon_bn254_sponge_caps(zp_ctx *ctx, uint64_t *h_state, const uint64_t *h_blocks, size_t nblocks, size_t extra,
                                      uint64_t *h_rates, uint64_t *h_caps);
size_t zp_merkle16_nodes(size_t M);
int32_t zp_merkle16_commit_bn254(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, uint64_t *d_tree);
int32_t zp_merkle16_open_bn254(zp_ctx *ctx, const uint64_t *d_tree, size_t M, size_t idx, uint64_t *h_path);
/* nq openings in one launch: h_paths u64[nq][levels][16][4] (levels = ceil(log16 M)) */
int32_t zp_merkle16_open_batch_bn254(zp_ctx *ctx, const uint64_t *d_tree, size_t M, const uint64_t *h_idx, int32_t nq, uint64_t *h_paths);

/* ---- Groth16 wrap, QAP step: NTT over the BN254 scalar field F_r (serves GenFinalProof, prover.proto:130-148 /
 * provider.rs:472-503; the G1/G2 multi-scalar multiplications are zp_msm_bn254* below).
 * Elements: 4 little-endian u64 words, standard form, < r.  Root of unity: 5^((r-1)/2^logn) (5 generates F_r^*, the
 * convention of the snarkjs/circom family), logn <= 28.
 * zp_ntt_bn254, in place, natural order in and out:
 *   inverse = 0: y_k = sum_i x_i (g w^k)^i          (h_coset = g as 4 words, or NULL for g = 1)
 *   inverse = 1: x_i = g^-i / N * sum_k y_k w^(-ik)  (the inverse of the above)
 * zp_qap_quotient_bn254: d_a, d_b, d_c hold the evaluations of A(x), B(x), C(x) on <w> (2^logm points each);
 *   on return d_a holds the 2^logm coefficients of H = (A B - C) / (x^m - 1) (the top one is 0 for a satisfied
 *   system); d_b, d_c are overwritten (evaluations on the coset g <w>).  g must not be an m-th root of unity.      */
int32_t zp_ntt_bn254(zp_ctx *ctx, uint64_t *d_data, int32_t logn, int32_t inverse, const uint64_t *h_coset);
int32_t zp_qap_quotient_bn254(zp_ctx *ctx, uint64_t *d_a, uint64_t *d_b, uint64_t *d_c, int32_t logm, const uint64_t *h_coset);

/* ---- the whole chunk STARK behind one call (serves GenChunkProof, prover.proto:56-66 / provider.rs:358-390) ----------
 * d_trace u64[W][2^logn] column-major in HBM, the statement as a constraint program blob (layout above), h_pubs its public
 * inputs.  Runs trace LDE + commitment, the stage-2 arguments of the program's table, the constraint quotient (in Q pieces),
 * out-of-domain evaluations, DEEP quotient, FRI, proof-of-work grinding and the query openings on the ctx's GPU; the
 * Fiat-Shamir transcript binds every parameter, the program's SHA-256 digest, the root of unity and the coset shift.
 * *out_json receives a malloc'ed, NUL-terminated proof text (*out_len bytes) to be released with zp_free_buffer; it is
 * byte-identical to what the Python orchestration (eigen_zeth_amd/stark/prover.py) writes for the same inputs and is what
 * oracle/stark_verify.py checks.  Conjectured security = n_queries * logb + pow_bits bits.  air_name only labels the proof.
 * trace_words = the number of u64 words behind d_trace; it must equal W * 2^logn with W from the program header (a short
 * buffer or a wrong logn is ZP_ERR_ARG instead of a read past the allocation).  PRECONDITION: trace values are canonical
 * (< p); they are not checked (a full pass over the witness) and non-canonical words give an unspecified, rejected proof.
 * On any error *out_json = NULL, *out_len = 0; no C++ exception leaves the call (ZP_ERR_NOMEM / ZP_ERR_INTERNAL).          */
int32_t zp_stark_prove(zp_ctx *ctx, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                       size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf,
                       int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, char **out_json, size_t *out_len);
/* the same in BN128-hash mode (the last STARK before the Groth16 wrap): 16-ary Poseidon-BN254 trees, transcript over the BN254
 * scalar field, no grinding (security = n_queries * logb); the t = 17 tables must be installed (zp_set_poseidon_bn254) */
int32_t zp_stark_prove_bn128(zp_ctx *ctx, const char *air_name, const uint64_t *h_program, size_t program_words, const uint64_t *d_trace,
                             size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn, int32_t logb, int32_t fri_logf,
                             int32_t fri_final_log, int32_t n_queries, char **out_json, size_t *out_len);
int32_t zp_free_buffer(void *p);
/* Hand the one-call provers of this ctx the GENERATED constraint kernel of a program (the AIR plug-in ABI below: `zpair_<air>_quotient` from the
 * shared library stark/air.py's code generator built; quotient_fn = its address, NULL forgets it): proofs of the program with this digest
 * evaluate their constraints through it instead of the interpreter -- same proof bytes, 0.7 instead of 1.2 ms at 2^21 x 76.  (Sparse periodic fixed
 * columns are read by the generated kernels since round 5; sharded proofs take the row-window form below since round 6.) */
int32_t zp_stark_set_air_kernel(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, void *quotient_fn);
/* (round 6) the same for the SHARDED provers of this ctx: `zpair_<air>_quotient_rows` of the same library evaluates a row window (explicit strides,
 * first row, b halo rows behind every column unless the window is the whole domain): int fn(stream, cols, stride_cols, fixed, stride_fixed, M, b,
 * row0, nrows, pub, apow, zhinv, xs_lo, xs_hi, lb, shift, wlast, out, stride_out).  Proof bytes are the same with the kernel or the interpreter. */
int32_t zp_stark_set_air_kernel_rows(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, void *quotient_rows_fn);
/* the AIR digest of a constraint program blob (host code, no ctx): SHA-256, out32 = the digest bytes (a proof's "air_digest" is the hex of the
 * first 8), out_words4 (or NULL) = the four 64-bit words (little-endian, mod p) a prover absorbs into its transcript and a verifier-AIR
 * witness builder needs for the inner proofs' statement (zp_recursion_witness: the head of the transcript stream). */
int32_t zp_program_digest(const uint64_t *h_program, size_t program_words, uint8_t *out32, uint64_t *out_words4);

/* ---- N5: FRI fold ------------------------------------------------------------------------------
 * d_in u64[3][2^logn] = f on shift*<w_n> (natural order);  d_out u64[3][2^(logn-logf)] =
 * sum_j beta^j g_j on shift^(2^logf)*<w_(n>>logf)>,  f(x) = sum_j x^j g_j(x^(2^logf)), logf in 1..4 */
int32_t zp_fri_fold(zp_ctx *ctx, const uint64_t *d_in, uint64_t *d_out, int32_t logn, int32_t logf,
                    const uint64_t beta[3], uint64_t shift);

/* ---- STARK stages around the committed columns (N4/N5 support) -------------------------------
 * zp_poly_eval_ext: p_c(z) for W polynomials with base-field coefficients u64[W][2^logn]
 *   (ascending) at the F_{p^3} point z; h_out[W][3].
 * zp_deep_quotient: on x = shift*w_M^r, r < 2^logm,
 *   F(x) = sum_{k<Wa+Wb} g^k (p_k(x)-e_k)/(x-z) + sum_{k<n_next} g^(Wa+Wb+k) (p_k(x)-e'_k)/(x-zw)
 *   p_k = columns of d_cols_a (k<Wa) then d_cols_b; e = h_ev_z[Wa+Wb][3], e' = h_ev_zw[n_next][3];
 *   d_out u64[3][2^logm].
 * zp_gather_rows: h_out[nq][W] = row h_idx[q] of the column-major matrix (query openings).
 * zp_merkle_open_batch: h_paths[nq][log2 M][4], bottom-up siblings for every queried leaf.        */
int32_t zp_poly_eval_ext(zp_ctx *ctx, const uint64_t *d_coef, int32_t logn, int32_t W, const uint64_t z[3],
                         uint64_t *h_out);
/* The verifier's side of a constraint program at the out-of-domain point (host code; csrc/verify.hip): the values of its K constraints
 * at zeta from the committed columns' evaluations -- what GenFinalProof's service checks natively on the aggregated proof a client hands
 * in (prover.proto:130-148) before it wraps it.  h_pubchal u64[n_pub + n_chal]: public inputs, then stage-2 challenge components;
 * h_ev_z / h_ev_zw u64[n_cols][3]; h_out u64[n_out][3] -- n_cols / n_out are the sizes of the CALLER's arrays and must equal the program's W + W2 / K (a
 * blob of another shape is refused, not read past them).  Fixed columns 0 / 1 = first-row / last-row selectors, then the sparse periodic
 * columns of the program (one shared inversion per column); threads <= 0: one per core (at most 16).  No ctx: nothing touches a GPU. */
int32_t zp_program_eval_ext(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn,
                            uint64_t root32, const uint64_t zeta[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, int32_t n_cols,
                            uint64_t *h_out, int32_t n_out, int32_t threads);
/* The FIXED columns of a program at the out-of-domain point: h_fixed u64[n_fixed][3] -- columns 0 / 1 the first-row / last-row selectors, then the
 * sparse periodic columns.  A function of (statement, public inputs, zeta) alone: the Groth16 wrap's circuit takes columns 2.. as committed inputs
 * (service/wrap_arith.py), a reader of its public input recomputes them.  Same arguments and refusals as zp_program_eval_ext.                       */
int32_t zp_program_fixed_eval_ext(const uint64_t *h_program, size_t program_words, const uint64_t *h_pubchal, int32_t n_pubchal, int32_t logn,
                                  uint64_t root32, const uint64_t zeta3[3], uint64_t *h_fixed, int32_t n_fixed, int32_t threads);
/* Out-of-domain evaluations FROM VALUES (barycentric form; round 5: the provers keep no coefficient buffers for this any more).
 * Column c is a polynomial p_c of degree < 2^logn given by its values on the coset shift*<w>, w of order 2^logn:
 *   d_cols[c * col_stride + i * row_stride] = p_c(shift * w^i)      (row_stride = 2^logb reads the 2^logn-point sub-coset of an
 *   extension committed on shift*<w_(2^(logn+logb))>; shift == 1 with row_stride == 1 reads a trace on its own domain;
 *   shift == 0 selects the ctx default).
 * h_ev_z[W][3] = p_c(z); with want_next != 0 also h_ev_zw[W][3] = p_c(z * w) (the same weights on the column rotated by one
 * row).  One pass over the columns.  ZP_ERR_ARG when z lies on the domain (a protocol excludes it; probability 2^-128).       */
int32_t zp_ood_eval(zp_ctx *ctx, const uint64_t *d_cols, size_t col_stride, size_t row_stride, int32_t W, int32_t logn,
                    uint64_t shift, const uint64_t z[3], int32_t want_next, uint64_t *h_ev_z, uint64_t *h_ev_zw);
int32_t zp_deep_quotient(zp_ctx *ctx, const uint64_t *d_cols_a, int32_t Wa, const uint64_t *d_cols_b, int32_t Wb,
                         int32_t logm, int32_t n_next, const uint64_t z[3], const uint64_t zw[3],
                         const uint64_t gamma[3], const uint64_t *h_ev_z, const uint64_t *h_ev_zw, uint64_t shift,
                         uint64_t *d_out);
/* the same for a window of the domain (one row shard of a proof spread over GPUs): local row j is row row0 + j of the
 * 2^logm-row domain; columns are given with explicit strides (elements), nrows rows each.                          */
int32_t zp_deep_quotient_rows(zp_ctx *ctx, const uint64_t *d_cols_a, int32_t Wa, size_t stride_a, const uint64_t *d_cols_b,
                              int32_t Wb, size_t stride_b, int32_t logm, size_t row0, size_t nrows, int32_t n_next,
                              const uint64_t z[3], const uint64_t zw[3], const uint64_t gamma[3], const uint64_t *h_ev_z,
                              const uint64_t *h_ev_zw, uint64_t shift, uint64_t *d_out, size_t stride_out);
/* grand product column of a permutation argument (stage-2 witness): Z[0]=1, Z[i+1]=Z[i]*(a[i]+g)/(b[i]+g)
 * in F_{p^3}; d_a, d_b u64[n]; d_out u64[3][n] plane-major.  Division by zero (b[i]+g = 0) is the
 * caller's concern (g is a Fiat-Shamir challenge).                                                   */
int32_t zp_grand_product(zp_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, size_t n, const uint64_t gamma[3],
                         uint64_t *d_out);
/* LogUp lookup columns (stage-2 witness of a lookup / range-check argument): values d_a u64[n] are looked up in the
 * table column d_t u64[n] with multiplicities d_m u64[n] (m[i] = how often t[i] is hit, counted once per table
 * value).  d_out u64[9][n] plane-major: h1 = 1/(a+g) (planes 0..2), h2 = m/(t+g) (3..5) and the running sum
 * S[0]=0, S[i+1]=S[i]+h1[i]-h2[i] (6..8), all in F_{p^3}.  The lookup holds iff S wraps to 0.           */
int32_t zp_logup_columns(zp_ctx *ctx, const uint64_t *d_a, const uint64_t *d_t, const uint64_t *d_m, size_t n,
                         const uint64_t gamma[3], uint64_t *d_out);
int32_t zp_gather_rows(zp_ctx *ctx, const uint64_t *d_cols, size_t M, int32_t W, const uint64_t *h_idx, int32_t nq,
                       uint64_t *h_out);
int32_t zp_merkle_open_batch(zp_ctx *ctx, const uint64_t *d_tree, size_t M, const uint64_t *h_idx, int32_t nq,
                             uint64_t *h_paths);

/* ---- AIR plug-in ABI (N4: row-parallel constraint evaluation + quotient) ---------------------------
 * Constraint kernels are generated per AIR (eigen_zeth_amd/stark/air.py -> generated/<air>.hip) and
 * built into libzpair_<air>.so, each exporting
 *   int zpair_<air>_quotient(void *hip_stream, const u64 *d_cols, const u64 *d_fixed, u64 M, u64 blowup,
 *        const u64 *d_pub, const u64 *d_alpha_pows, const u64 *d_zhinv, const u64 *d_xs_lo,
 *        const u64 *d_xs_hi, int lb, u64 shift, u64 w_last, u64 *d_out);
 *   d_cols [W][M] LDE of the trace, d_fixed [2][M] LDE of L_first/L_last, d_alpha_pows [K][3],
 *   d_zhinv [blowup] = 1/Z_H on the coset (periodic), d_out [3][M] = quotient planes.
 *   and (round 6) the same kernel for a ROW WINDOW of a sharded proof:
 *   int zpair_<air>_quotient_rows(void *hip_stream, const u64 *d_cols, u64 stride_cols, const u64 *d_fixed, u64 stride_fixed, u64 M, u64 blowup,
 *        u64 row0, u64 nrows, const u64 *d_pub, const u64 *d_alpha_pows, const u64 *d_zhinv, const u64 *d_xs_lo, const u64 *d_xs_hi, int lb,
 *        u64 shift, u64 w_last, u64 *d_out, u64 stride_out);
 *   rows [row0, row0 + nrows) of the M-row domain; unless nrows == M every column carries the window's blowup halo rows behind it.
 * zp_domain_tables hands such kernels the ctx-owned two-level table of w_M^e (x = shift*lo[e&mask]*hi[e>>lb]). */
int32_t zp_domain_tables(zp_ctx *ctx, int32_t logm, const uint64_t **d_lo, const uint64_t **d_hi, int32_t *lb);
/* ---- N4 as data: the constraint program ----------------------------------------------------------------
 * An AIR handed over as a u64 blob instead of a generated kernel (a host without hipcc at run time can still
 * prove any AIR; slower than the plug-in kernel, same results).  Layout, all words u64 little-endian:
 *   [0] magic "ZPAIR1\0\0"   [1] trace width W   [2] stage-2 width W2   [3] fixed columns (2: L_first, L_last)
 *   [4] publics   [5] challenges (0 or 3; they follow the publics in the operand space)   [6] n_const   [7] n_instr
 *   [8] n_constraints   [9] n_slots   [10] n_stage2   [11] quotient chunks Q (pieces of degree < N)
 *   consts[n_const] | instr[n_instr] | stage2[n_stage2][4] = {kind 1 perm: a, b, 0 | kind 2 lookup: a, t, m}
 *   instr: bits 0-7 opcode (1 add, 2 sub, 3 mul: slot[dst] = a op b;  4 OUT: constraint k = a, k counts the OUTs),
 *          8-23 dst slot, 24-27 kind(a), 28-43 index(a), 44-47 kind(b), 48-63 index(b)
 *   operand kinds: 0 slot, 1 column at row r, 2 column at the next trace row (r + blow-up), 3 fixed column,
 *                  4 public/challenge, 5 constant, 6 the factor x - w_N^(N-1) of a transition constraint
 * zp_eval_quotient evaluates the program on every row of the LDE domain (M = 2^logm rows, x = shift*w_M^r), combines
 * constraint k with alpha^k in F_{p^3} (h_alpha_pows [K][3]), multiplies by h_zhinv[r mod 2^logb] and writes the three
 * planes d_out u64[3][M].  d_cols u64[W+W2][M], d_fixed u64[2][M]; h_pub = publics then challenges (n_pub values).
 * A malformed program (bad magic, lengths, operand or slot out of range, more than 24 slots) is ZP_ERR_ARG.        */
int32_t zp_eval_quotient(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *d_cols,
                         const uint64_t *d_fixed, int32_t logm, int32_t logb, const uint64_t *h_pub, int32_t n_pub,
                         const uint64_t *h_alpha_pows, const uint64_t *h_zhinv, uint64_t shift, uint64_t w_last,
                         uint64_t *d_out);
/* ---- fixed columns of a statement on the evaluation domain: the d_fixed of zp_eval_quotient -------------------------
 * Fixed columns 0 and 1 are the boundary selectors L_first, L_last.  Columns 2 .. n_fixed-1 (program header word 3) are
 * SPARSE PERIODIC columns the verifier knows: a table behind the stage-2 table holds, per column, one word
 * [lp | n_entries << 8] and n_entries pairs (pos | is_pub << 63, value): the column has period 2^lp <= N (lp = logn: not
 * periodic), is zero except at rows pos + k 2^lp, where it holds the constant `value` or, with is_pub, public input number
 * `value`.  (Round constants, schedule selectors and per-row expected values of a verifier AIR are such columns.)
 * zp_fixed_columns fills d_out (>= zp_fixed_columns_words() words; 0 = malformed program / lp > logn) with
 *   [L_first: 2^(logn+logb)][L_last: 2^(logn+logb)][column 2: 2^(lp_2+logb)][column 3: ...]
 * i.e. the two selectors extended to the whole coset and ONE extended period of every periodic column (a period-p column is
 * g(x^(N/p)); on shift*<w_M> that is the extension of p values with coset shift shift^(N/p), read at row mod p*2^logb).   */
size_t zp_fixed_columns_words(const uint64_t *h_program, size_t program_words, int32_t logn, int32_t logb);
int32_t zp_fixed_columns(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub,
                         int32_t logn, int32_t logb, uint64_t shift, uint64_t *d_out, size_t out_words);
/* the same for a window [row0, row0 + nrows) of the domain (a row shard): columns with explicit strides; unless the
 * window is the whole domain the caller appends the 2^logb halo rows (rows row0+nrows .. of the domain, wrapping to 0)
 * behind each column (stride_cols >= nrows + 2^logb); row0 and nrows are multiples of 2^logb.                      */
int32_t zp_eval_quotient_rows(zp_ctx *ctx, const uint64_t *h_program, size_t program_words, const uint64_t *d_cols,
                              size_t stride_cols, const uint64_t *d_fixed, size_t stride_fixed, int32_t logm, int32_t logb,
                              size_t row0, size_t nrows, const uint64_t *h_pub, int32_t n_pub, const uint64_t *h_alpha_pows,
                              const uint64_t *h_zhinv, uint64_t shift, uint64_t w_last, uint64_t *d_out, size_t stride_out);
/* synthetic witness generation (stands in for the zkVM executor, which is not obtainable offline):
 * kind 0 = Fibonacci (W=2), kind 1 = wide degree-2 mix (any W >= 3), kind 2 = permutation AIR (W=3:
 * a, b = a permuted, c = a^2), kind 3 = chunk AIR (W >= 12: W-8 wide-mix columns, then Fibonacci a,b, range values r,
 * r permuted, range table t, multiplicities m, r^2, a*r+b).  h_trace u64[W][2^logn]; h_pub (room for 8) receives the
 * public inputs (3 / min(4,W) / 1 / 8).                 */
int32_t zp_synth_trace(int32_t kind, int32_t logn, int32_t W, uint64_t seed, uint64_t *h_trace, uint64_t *h_pub);
/* the same generators with the first n_bind starting values dictated by the caller (canonical, < p) instead of drawn from
 * the seed; the AIRs constrain exactly those cells to public inputs, so the proof names them.  The service puts the limbs
 * of the block statement there (pre/post state root, transaction digest: what GenBatchChunksResult reports,
 * proto/prover/v1/prover.proto:80-91, consumed at src/prover/provider.rs:315-330), which binds a chunk proof to its block.
 * kind 0: bind[0..1] = a[0], b[0];  kind 1: bind[i] = c_i[0], i < min(4, W);  kind 3: bind[0..3] = c_0..c_3[0],
 * bind[4..5] = Fibonacci a[0], b[0].  n_bind above that (or kind 2) is ZP_ERR_ARG.                                   */
int32_t zp_synth_trace_bound(int32_t kind, int32_t logn, int32_t W, uint64_t seed, const uint64_t *bind, int32_t n_bind,
                             uint64_t *h_trace, uint64_t *h_pub);

/* the same traces generated IN HBM (csrc/synth.hip), word for word what zp_synth_trace_bound writes: for batches whose host generator
 * would bound the prover (64 chunks of 2^22 x 76: 60 CPU-seconds).  The wide-mix columns are a recurrence over the rows, so a trace is
 * filled from CHECKPOINTS of that recurrence (the row at every 4096-th position): zp_synth_checkpoints walks the recurrences of n_chunks
 * chunks at once, one wave per chunk (h_seeds[n_chunks], h_bind[n_chunks][n_bind]; d_ckpt holds n_chunks x zp_synth_checkpoint_words
 * words, chunk after chunk; synchronous), zp_synth_trace_device(.., d_ckpt of that chunk, ..) fills d_trace u64[W][2^logn] from them on
 * the ctx stream (asynchronous; h_pub is written before it returns).  d_ckpt = NULL: the checkpoints of the one chunk are made inside the
 * call (kinds 0 and 2 have no recurrence column and need none).  kind 1 / 3: at most 256 wide-mix columns.                            */
size_t zp_synth_checkpoint_words(int32_t kind, int32_t logn, int32_t W);
int32_t zp_synth_checkpoints(zp_ctx *ctx, int32_t kind, int32_t logn, int32_t W, int32_t n_chunks, const uint64_t *h_seeds,
                             const uint64_t *h_bind, int32_t n_bind, uint64_t *d_ckpt);
int32_t zp_synth_trace_device(zp_ctx *ctx, int32_t kind, int32_t logn, int32_t W, uint64_t seed, const uint64_t *h_bind, int32_t n_bind,
                              const uint64_t *d_ckpt, uint64_t *d_trace, uint64_t *h_pub);

/* synthetic MSM input (stands in for a proving key, which the offline build cannot obtain): n DISTINCT points
 * P_i = (start + i) * G of BN254 G1 in the zp_msm_bn254 layout, generated on the host with `threads` threads (0 = all);
 * start must exceed 1024.  Known discrete logs: sum_i s_i P_i = (sum_i s_i (start + i) mod r) * G.            */
int32_t zp_synth_g1_points(uint64_t start, size_t n, uint32_t *h_points, int32_t threads);

/* ---- recursive-proof TEXT -> arrays (GenAggregatedProof / GenFinalProof take their inner proofs as strings: proto/prover/v1/prover.proto:
 * 115-148, sent verbatim by src/prover/provider.rs:422-433,472-483).  Host code, no ctx.  97 % of a proof text is the decimal numbers of its
 * query openings: zp_proof_queries_scan finds the "queries" array of a proof object and sizes it (queries, stage-2 tree or not, FRI layers,
 * per tree -- trace, [stage2], quotient, fri0 .. -- leaf width and path length; every query alike), zp_proof_queries_parse writes index
 * u64[nq], values (per tree a block u64[nq][w_t]) and paths (per tree a block u64[nq][d_t][4]).  The caller reads the rest of the text (the
 * header: kilobytes) with its own JSON parser after cutting [q_begin, q_end) out.  zp_json_key_span: the byte span of the value of a member of
 * the top-level object (the "stark" of an aggregated proof).  Strict: any other grammar is ZP_ERR_ARG (the caller falls back to its parser). */
int32_t zp_json_key_span(const char *text, size_t len, const char *key, size_t *begin, size_t *end);
int32_t zp_proof_queries_scan(const char *text, size_t len, size_t *q_begin, size_t *q_end, int32_t *n_queries, int32_t *has_stage2, int32_t *n_fri,
                              int32_t *widths, int32_t *depths, int32_t max_trees);
int32_t zp_proof_queries_parse(const char *text, size_t q_begin, size_t q_end, int32_t n_queries, int32_t has_stage2, int32_t n_fri,
                               const int32_t *widths, const int32_t *depths, uint64_t *index, uint64_t *values, uint64_t *paths);

/* ---- witness of the STARK-verifier AIR: its arithmetic columns (GenAggregatedProof / the final STARK of GenFinalProof:
 * proto/prover/v1/prover.proto:115-148, src/prover/provider.rs:422-503; the AIR is eigen_zeth_amd/stark/verifier_air.py) -----------------
 * Besides the permutation blocks (zp_poseidon_trace) the verifier trace has 21 columns that carry the field arithmetic of a verifier at
 * the queries of its inner proofs: registers copying the opened values, the Horner accumulator of the DEEP sums, the evaluation points
 * spelled by the path bits, the interpolation / fold accumulators of every FRI layer.  desc: the schedule of the AIR as data (u64 words:
 * header, one word per block of a period, one record per committed tree, the inverse-DFT tables of the folds; built by
 * verifier_air.arith_descriptor).  vals u64[openings][max_w]: the opened values; index u64[openings]: leaf indices; dbit u64[blocks]:
 * direction bit of every block; blk_op i64[blocks]: block -> opening (-1: none); arith_pubs u64[proofs][ap_n], final_vals
 * u64[slots][proofs][3]: the arithmetic public inputs.  Output u64[21][32 * blocks].  Returns -10 / -11 when the opened values are
 * inconsistent (a FRI layer does not hold the value the layer before claims / the last fold is not the final layer): such inner proofs
 * have no accepting witness.  zp_verifier_arith_host needs no GPU (the walk is host code either way; periods run on `threads` threads,
 * 0 = all); zp_verifier_arith_trace expands the per-block records in HBM.                                                              */
int32_t zp_verifier_arith_host(const uint64_t *desc, size_t desc_words, const uint64_t *vals, const uint64_t *index, const uint64_t *dbit,
                               const int64_t *blk_op, const uint64_t *arith_pubs, const uint64_t *final_vals, uint64_t *h_out, int32_t threads);
int32_t zp_verifier_arith_trace(zp_ctx *ctx, const uint64_t *desc, size_t desc_words, const uint64_t *vals, const uint64_t *index,
                                const uint64_t *dbit, const int64_t *blk_op, const uint64_t *arith_pubs, const uint64_t *final_vals,
                                uint64_t *d_out, int32_t threads);

/* The WHOLE witness of the verifier AIR behind one call -- what a host does for GenAggregatedProof / the final STARK between parsing the
 * inner proofs and zp_stark_prove(_bn128): every opening hashed (leaf sponge + path) level-synchronously on the GPU, the inner Fiat-Shamir
 * transcripts replayed (zp_poseidon_sponge_caps), the public inputs assembled, the permutation blocks traced (zp_poseidon_trace), the
 * arithmetic columns walked and expanded: d_trace u64[47][32 * blocks] is assembled in HBM, h_pubs receives the public inputs
 * (zp_recursion_publics_words(desc) words: roots | leaf indices | transcripts | arithmetic constants | final-layer values).
 * desc: verifier_air.arith_descriptor(shape).  Per inner proof p: h_index[p] u64[nq], h_values[p] / h_paths[p] in the layout of
 * zp_proof_queries_parse (trees in the order trace, [stage2], quotient, fri0 ..), h_stream[p] = what its transcript absorbs, in protocol
 * order: AIR digest words[4] | public inputs | roots of trace, [stage2], quotient (4 each) | evaluations at zeta ((W + W2 + Wq) x 3) | at
 * zeta w ((W + W2) x 3) | FRI roots (4 each) | final layer (3 planes) | grinding nonce (if the proofs grind).
 * Returns -13: an opening does not hash to its root; -14: the transcript does not give the proof's query indices / the nonce fails;
 * -10 / -11: the opened values are inconsistent (zp_verifier_arith_host) -- inner proofs that do not verify have no accepting witness. */
int32_t zp_recursion_witness(zp_ctx *ctx, const uint64_t *desc, size_t desc_words, const uint64_t *const *h_index, const uint64_t *const *h_values,
                             const uint64_t *const *h_paths, const uint64_t *const *h_stream, const size_t *stream_words, uint64_t *d_trace,
                             uint64_t *h_pubs, size_t pubs_words, int32_t threads);
size_t zp_recursion_publics_words(const uint64_t *desc, size_t desc_words);

/* ---- R1CS over the BN254 scalar field for the Groth16 wrap (GenFinalProof, prover.proto:130-148; consumer ethereum/mod.rs:338-394) -------
 * Host code, no ctx.  A circuit = many instances of ONE gadget template (a width-17 Poseidon-BN254 permutation: three sparse matrices over
 * local wires) + explicit extra constraints; blob layout in eigen_zeth_amd/service/r1cs.py.  zp_r1cs_eval completes the witness (the internal
 * wires of every instance, in order), checks EVERY constraint and writes A w, B w, C w (u64[2^logm][4], standard form: the input of
 * zp_qap_quotient_bn254); -20: the assignment does not satisfy the circuit (*bad = first violated constraint); -21: a wire nobody set.
 * zp_r1cs_key_scalars: the scalars of a Groth16 key at tau (params = tau, alpha, beta, gamma, delta; a LOCAL seeded setup, not a ceremony):
 * u_j(tau), v_j(tau), l_j = (beta u_j + alpha v_j + w_j) / delta (or / gamma for the constant and the public inputs), h_i = tau^i Z(tau) / delta;
 * the group elements are these times the generators: zp_fixed_base_mul_bn254 / _g2.                                                        */
int32_t zp_r1cs_eval(const uint64_t *circ, size_t words, uint64_t *witness, uint8_t *set, uint64_t *a_ev, uint64_t *b_ev, uint64_t *c_ev, int64_t *bad);
int32_t zp_r1cs_key_scalars(const uint64_t *circ, size_t words, const uint64_t *params, uint64_t *out_u, uint64_t *out_v, uint64_t *out_l, uint64_t *out_h,
                            int32_t threads);

/* ---- the Groth16 wrap behind two calls (GenFinalProof: prover.proto:130-148, src/prover/provider.rs:472-503; the consumer of the result is
 * src/settlement/ethereum/mod.rs:338-394) ------------------------------------------------------------------------------------------------------
 * zp_stark_openings: roots, query indices, opened values and 16-ary paths of the LAST zp_stark_prove_bn128 proof of this ctx, in binary (what its
 *   text carries; layout at the definition, csrc/prove.hip); *out points into the ctx until the next proof on it.
 * zp_wrap_assign: the caller-set wires of the wrap circuit (service/wrap_circuit.py) from those openings, driven by the assignment script the
 *   circuit builder writes beside the circuit blob; aux4 = the extra field element the proof is bound to (the aggregator address), standard form.
 *   out_idx u64[cap], out_val u64[cap][4]; *n_set = wires written.
 * zp_r1cs_eval_device: zp_r1cs_eval on the GPU -- the n_set caller-set wires in (host), the complete witness d_w u64[n_wires][4] and A w, B w, C w
 *   (d_a, d_b, d_c u64[2^logm][4]) out in HBM, the public inputs out_pub u64[n_pub][4] to the host; an instance's wires and rows are the
 *   intermediate values of its permutation (one kernel per wave of instances), the explicit constraints go through a sparse-row kernel.  Same
 *   results and refusals (-20 / -21, *bad) as zp_r1cs_eval.  The circuit's gadget must be the width-17 Poseidon permutation of the installed tables:
 *   it is compared with the kernel on one instance, once per circuit and ctx (ZP_ERR_ARG otherwise).
 * zp_groth16_prove: witness completion + A w, B w, C w (zp_r1cs_eval_device), the QAP quotient, five MSMs over the key's device-resident points
 *   (d_u1x: u32[n_wires + 2][16] = [u_j]_1 | alpha_1 | delta_1; d_v_wires u32[n_v]: the wires with a non-zero column in B, ascending -- the B side of
 *   the key holds only those: d_v1x u32[n_v + 2][16] = [v_j]_1 | beta_1 | delta_1, d_v2x u32[n_v + 2][32] = [v_j]_2 | beta_2 | delta_2; d_l1: u32[n_wires][16], infinity at the constant and the public inputs; d_h1: u32[2^logm - 1][16]; h_delta1 u32[16]) and the blinding
 *   terms for (h_r, h_s).  out_a u32[16], out_b u32[32], out_c u32[16] = pi_a, pi_b, pi_c; out_pub u64[n_pub][4]; h_ms (may be NULL) double[8] =
 *   milliseconds of witness, QAP, all MSMs, then A, B (G1), B (G2), l, h one by one.  -20 / -21 as zp_r1cs_eval: a false statement has no proof.
 * zp_sha256: SHA-256 of a byte string (the digests proof texts name; the deterministic blinding of a test run).                                    */
int32_t zp_stark_openings(zp_ctx *ctx, const uint64_t **out, size_t *words);
/* (round 6) aux u64[n_aux][4]: element 0 the value the proof is bound to (the aggregator address), elements 1.. what else the circuit takes from its
 * caller -- wrap stage B-2: the statement's sparse fixed columns at zeta (zp_program_fixed_eval_ext, columns 2..), one element per column, the three
 * components packed c0 + c1 2^64 + c2 2^128.  The openings record is "PZOPEN03": it ends with the rate element behind every challenge.              */
int32_t zp_wrap_assign(const uint64_t *script, size_t script_words, const uint64_t *openings, size_t open_words, const uint64_t *aux, size_t n_aux,
                       uint64_t *out_idx, uint64_t *out_val, size_t cap, size_t *n_set);
/* That aux list from what the host of GenFinalProof holds: the openings record (challenge element 1 is zeta), the final STARK's statement (program,
 * public inputs, log2 of the trace length, root of unity) and the element the proof is bound to (addr4).  out_aux u64[cap][4], *n_aux = the program's
 * n_fixed - 1; zeta3_out (may be NULL) u64[3] = zeta (what the proof text carries so that a reader can recompute the fixed columns).               */
int32_t zp_wrap_aux(const uint64_t *openings, size_t open_words, const uint64_t *h_program, size_t program_words, const uint64_t *h_pub, int32_t n_pub,
                    int32_t logn, uint64_t root32, const uint64_t *addr4, uint64_t *out_aux, size_t cap, size_t *n_aux, uint64_t *zeta3_out);
int32_t zp_r1cs_eval_device(zp_ctx *ctx, const uint64_t *circ, size_t words, const uint64_t *set_idx, const uint64_t *set_val, size_t n_set, uint64_t *d_w,
                            uint64_t *d_a, uint64_t *d_b, uint64_t *d_c, uint64_t *out_pub, int64_t *bad);
int32_t zp_groth16_prove(zp_ctx *ctx, const uint64_t *circ, size_t words, const uint32_t *d_u1x, const uint32_t *d_v_wires, size_t n_v, const uint32_t *d_v1x,
                         const uint32_t *d_v2x, const uint32_t *d_l1, const uint32_t *d_h1, const uint32_t *h_delta1, const uint64_t *set_idx, const uint64_t *set_val, size_t n_set,
                         const uint64_t *h_r, const uint64_t *h_s, uint32_t *out_a, uint32_t *out_b, uint32_t *out_c, uint64_t *out_pub, double *h_ms,
                         int64_t *bad);
int32_t zp_sha256(const uint8_t *data, size_t len, uint8_t *out32);

/* ---- multi-GPU: RCCL over xGMI behind the C-ABI (BASELINE.json configs[3]) ---------------------------------
 * One process per GPU; a zp_comm joins this rank's ctx to the RCCL communicator of `world` ranks (a power of two).  Rank 0 makes
 * the 128-byte id (zp_comm_unique_id) and hands it to the others out of band (a file, the service's own channel); every rank
 * then calls zp_comm_create -- collectively.  All collectives run on the ctx stream; librccl.so is loaded at run time.
 *   zp_comm_all_to_all : recv[h * w ..) <- rank h's send[me * w ..)   (grouped ncclSend/ncclRecv: each pair on its own link)
 *   zp_comm_all_gather : recv[h * w ..) <- rank h's send[0 .. w)
 *   zp_exchange_columns_to_rows : this rank's columns u64[Wl][M] -> ALL columns of this rank's rows u64[G*Wl][M/G]
 *                                 (zp_pack_blocks + one all-to-all; d_pack = scratch of Wl*M words; the result is the
 *                                 column-major matrix zp_merkle_commit takes)
 *   zp_merkle_commit_sharded    : ONE Poseidon Merkle commitment over the ranks: columns in, the global root out (equal to the
 *                                 single-GPU root over all G*Wl columns); d_tree_local = this rank's subtree over its M/G rows,
 *                                 (2M/G - 1) * 4 words; the G sub-roots are all-gathered, the top log2 G levels hashed on
 *                                 every rank.  Constraint / DEEP stages of a sharded proof take row windows:
 *                                 zp_eval_quotient_rows, zp_deep_quotient_rows.
 *   zp_comm_all_reduce_sum      : buf <- sum over the ranks of buf (64-bit wrapping sums)
 * A communicator WITHOUT RCCL for ranks that live in ONE process (threads, one ctx each, on one GPU or several): create a
 * group (zp_comm_group_create(world)), then every rank's zp_comm_create_local(ctx, rank, group); its collectives are device-to-
 * device copies around a thread barrier.  RCCL refuses two ranks on one device, so this is how the multi-rank logic of the sharded
 * entry points is exercised on a one-GPU box; it is also a transport for a single-process multi-GPU host.
 *   zp_stark_prove_sharded      : ONE chunk STARK over the ranks of a communicator: every rank passes ITS trace columns: with
 *                                 wl = ceil(W / world), rank r owns columns [r wl, min((r+1) wl, W)) (d_trace_local u64[its
 *                                 columns][2^logn], trace_words = columns << logn; W need not divide: the tail ranks hold fewer
 *                                 columns, or none), all other arguments as zp_stark_prove, identical on every rank;
 *                                 every rank receives the same proof text, byte for byte what zp_stark_prove writes for the whole
 *                                 trace on one GPU.  Column-sharded: LDE, out-of-domain evaluations; row-sharded: the three
 *                                 commitments, the constraint quotient (with a blow-up halo), the DEEP quotient; replicated: the
 *                                 stage-2 columns, FRI.  Goldilocks-hash mode.  Collective: every rank must call it.
 *   zp_stark_prove_sharded_bn128: the same in BN128-hash mode (the last STARK before the Groth16 wrap, src/prover/provider.rs:431-
 *                                 451): zp_stark_prove_bn128's text, byte for byte, on every rank, and its openings record
 *                                 (zp_stark_openings) on every rank's ctx.  The 16-ary trace tree is sharded by rows (local
 *                                 sub-trees of 16^h c rows, ONE all-gather of their c top digests, the levels above replicated),
 *                                 which needs one row per leaf: more than 28 trace columns (the 47-column verifier AIR); the
 *                                 narrow stage-2 and quotient trees, whose leaves hold rows of every shard, are built replicated
 *                                 from columns every rank holds whole anyway.  zp_set_poseidon_bn254(ctx, 17, ..) on every ctx.
 * FAILURE: no rank waits for ever.  A rank whose step fails inside a collective or a sharded entry point takes the communicator down
 * before it returns its own error; its peers return ZP_ERR_COMM (in-process group: at once, woken from the barrier; RCCL: when their
 * watchdog expires and aborts the communicator, ncclCommAbort).  A host whose rank fails BETWEEN collectives calls zp_comm_abort.  A
 * dead communicator answers every later collective with ZP_ERR_COMM; destroy it and build a new one.  zp_comm_set_timeout_ms: how long
 * a collective waits for its peers (default 120 000; RCCL: 0 = asynchronous collectives without a watchdog).                         */
typedef struct zp_comm zp_comm;
typedef struct zp_comm_group zp_comm_group;
int32_t zp_comm_unique_id(uint8_t *out128);
int32_t zp_comm_create(zp_ctx *ctx, int32_t rank, int32_t world, const uint8_t *id128, zp_comm **out);
int32_t zp_comm_destroy(zp_comm *comm);
int32_t zp_comm_rank(const zp_comm *comm);
int32_t zp_comm_world(const zp_comm *comm);
/* What the TRANSPORT says about the communicator (not what its creator was told): out4 = {transport: 1 RCCL / 0 in-process group, ranks the
 * communicator joined (ncclCommCount), this rank inside it (ncclCommUserRank), the HIP device it is bound to (ncclCommCuDevice); -1 where the
 * loaded RCCL lacks the query}.  bench.py prints it ("rccl"): a launcher that started N worlds of one rank instead of one world of N shows here. */
int32_t zp_comm_info(const zp_comm *comm, int32_t *out4);
/* zp_merkle_commit_sharded keeps its exchange buffers (pack, rows) in the communicator between calls, grown to the largest matrix committed;
 * this gives them back to the device (zp_comm_destroy does too). */
int32_t zp_comm_release_scratch(zp_comm *comm);
int32_t zp_comm_abort(zp_comm *comm);
int32_t zp_comm_set_timeout_ms(zp_comm *comm, int32_t ms);
int32_t zp_comm_all_to_all(zp_comm *comm, const uint64_t *d_send, uint64_t *d_recv, size_t words_per_peer);
int32_t zp_comm_all_gather(zp_comm *comm, const uint64_t *d_send, uint64_t *d_recv, size_t words);
int32_t zp_comm_broadcast(zp_comm *comm, uint64_t *d_buf, size_t words, int32_t root);
int32_t zp_comm_all_reduce_sum(zp_comm *comm, uint64_t *d_buf, size_t words);
int32_t zp_comm_group_create(int32_t world, zp_comm_group **out);
int32_t zp_comm_group_destroy(zp_comm_group *group);
int32_t zp_comm_create_local(zp_ctx *ctx, int32_t rank, zp_comm_group *group, zp_comm **out);
int32_t zp_stark_prove_sharded(zp_comm *comm, const char *air_name, const uint64_t *h_program, size_t program_words,
                               const uint64_t *d_trace_local, size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn,
                               int32_t logb, int32_t fri_logf, int32_t fri_final_log, int32_t n_queries, int32_t pow_bits, char **out_json,
                               size_t *out_len);
int32_t zp_stark_prove_sharded_bn128(zp_comm *comm, const char *air_name, const uint64_t *h_program, size_t program_words,
                                     const uint64_t *d_trace_local, size_t trace_words, const uint64_t *h_pubs, int32_t n_pubs, int32_t logn,
                                     int32_t logb, int32_t fri_logf, int32_t fri_final_log, int32_t n_queries, char **out_json, size_t *out_len);
int32_t zp_exchange_columns_to_rows(zp_comm *comm, const uint64_t *d_cols, size_t Wl, size_t M, uint64_t *d_pack, uint64_t *d_rows);
/* four-step NTT of ONE column of 2^logn elements split over the ranks (BASELINE.json configs[3]: "RCCL all-to-all over xGMI for the
 * four-step NTT transpose"; replaces the torch.distributed orchestration eigen_zeth_amd/multigpu.py:four_step_ntt): d_data = this
 * rank's contiguous block of 2^logn / G elements, transformed in place (natural_output != 0: this rank's contiguous block of the
 * transform, three all-to-alls; 0: rows k1 of Y[k1][k2] = X[k1 + N1 k2], two all-to-alls); d_tmp = scratch of 2 * 2^logn / G words;
 * inverse != 0: inverse transform incl. 1/N.  Bit-identical to zp_ntt / zp_intt of the whole column on one GPU. */
int32_t zp_ntt_sharded(zp_comm *comm, uint64_t *d_data, uint64_t *d_tmp, int32_t logn, int32_t inverse, int32_t natural_output);
int32_t zp_merkle_commit_sharded(zp_comm *comm, const uint64_t *d_cols, size_t M, int32_t Wl, uint64_t *d_tree_local, uint64_t *h_root4);

/* ---- N6: BN254 (alt_bn128) G1 multi-scalar multiplication ---------------------------------------
 * d_points u32[n][16]: affine x (8 little-endian 32-bit limbs) then y, standard (non-Montgomery)
 * integers < q; (0,0) encodes the point at infinity.  d_scalars u32[n][8] little-endian, any 256-bit
 * value (used mod the group order implicitly).  h_out u32[16] = affine sum, all zero = infinity.
 * Pippenger; inputs above 2^24 points run as 2^24-point passes; scratch (about 300 B per point of a pass) comes from
 * an arena the ctx keeps until zp_destroy.  Skewed scalars (many equal/small values) are handled in parallel.   */
int32_t zp_msm_bn254(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n, uint32_t *h_out);
/* the same over G2 (the B element of a Groth16 proof): affine points over F_q2 = F_q[u]/(u^2+1) on the twist
 * y^2 = x^3 + 3/(9+u); d_points u32[n][32] = x.c0, x.c1, y.c0, y.c1 (8 little-endian words each, standard form,
 * all-zero = infinity), d_scalars u32[n][8]; h_out u32[32] in the same layout (all-zero = infinity).          */
int32_t zp_msm_bn254_g2(zp_ctx *ctx, const uint32_t *d_points, const uint32_t *d_scalars, size_t n,
                        uint32_t *h_out);

/* fixed-base multiplication: h_points[i] = h_scalars[i] * B for ONE base point B (h_base: affine, the layout of an MSM point; G1: u32[16], G2:
 * u32[32]) -- the group elements of a Groth16 key from the scalars of zp_r1cs_key_scalars (an MSM sums, this does not).  Host buffers in and
 * out (h_scalars u32[n][8], h_points in the MSM layout, (0, 0) = infinity); 32 table additions per scalar on the GPU, the Jacobian results made
 * affine on `threads` host threads (0 = all) with batched inversions.  Key-generation work: once per key, not per proof.                  */
int32_t zp_fixed_base_mul_bn254(zp_ctx *ctx, const uint32_t *h_base, const uint32_t *h_scalars, size_t n, uint32_t *h_points, int32_t threads);
int32_t zp_fixed_base_mul_bn254_g2(zp_ctx *ctx, const uint32_t *h_base, const uint32_t *h_scalars, size_t n, uint32_t *h_points, int32_t threads);

/* ---- host-buffer conveniences (H2D + compute + D2H + sync), the form a non-GPU-aware host uses */
int32_t zp_ntt_host(zp_ctx *ctx, uint64_t *h_cols, int32_t logn, int32_t W, int32_t inverse);
int32_t zp_lde_host(zp_ctx *ctx, const uint64_t *h_in, uint64_t *h_out, int32_t logn, int32_t logb,
                    int32_t W, uint64_t shift);
int32_t zp_merkle_commit_host(zp_ctx *ctx, const uint64_t *h_cols, size_t M, int32_t W, uint64_t *h_tree);

/* ---- measurement ---------------------------------------------------------------------------
 * With profiling on, every NTT pass launch of the next transforms is bracketed by HIP events on the
 * ctx stream.  zp_get_pass_timings synchronises, then returns for the launches recorded since the
 * last call: duration in ms and log2(radix) (negative for the transposing first pass).           */
int32_t zp_set_profiling(zp_ctx *ctx, int32_t on);
int32_t zp_get_pass_timings(zp_ctx *ctx, float *ms, int32_t *radix_log, int32_t cap, int32_t *count);
/* JSON array [{"stage": name, "ms": t}, ...] of the compute entry points called since the last call
 * (device time between HIP events on the ctx stream; profiling must be on).                          */
int32_t zp_stage_timings(zp_ctx *ctx, char *buf, size_t buflen);

/* device-to-device copy of `bytes` (multiple of 16) with the library's own 16-byte-per-lane non-temporal kernel, `reps`
 * times after one warm-up; *ms_per_copy = average duration (HIP events).  The HBM ceiling bench.py reports.      */
int32_t zp_hbm_copy_probe(zp_ctx *ctx, const void *d_src, void *d_dst, size_t bytes, int32_t reps, float *ms_per_copy);
/* experiment knobs for kernel tuning sweeps (keys: "ntt_logt" 4|5 tile of the radix-256 pass, "ntt_logt9" 4|5, "ntt_tpw" tiles per workgroup,
 * "ntt_chunk_log" log2 of the elements per NTT launch (27), "ntt_maxl" largest log2 radix of a pass (9; 10..12 select the 1024-thread two-pass plans), "merkle_coop_log" largest tree level given to the 12-lanes-per-node
 * kernel (15), "msm_chunk_log" log2 of the points per Pippenger run (24), "msm_c" window width, "ntt_small_wave" / "fri_fold_lanes" the in-wave (DPP / ds_swizzle) forms of the
 * small transform and of the fold by 16: 0 where they measured faster, 1 always, 2 never; "ntt_logt12" 1: radix-4096 passes on 64-KiB tiles, two
 * 512-thread workgroups per CU (2: 128-KiB tiles); "p254_block" 2: the lane-per-permutation Poseidon-BN254 kernel walks its partial rounds one by
 * one instead of in blocks of four; 0 = default); not for production hosts */
int32_t zp_set_tuning(zp_ctx *ctx, const char *key, int32_t value);

/* ---- introspection ------------------------------------------------------------------------- */
/* JSON description of the pass plan used for a 2^logn transform (for DESIGN/bench reporting) */
int32_t zp_ntt_plan_json(zp_ctx *ctx, int32_t logn, char *buf, size_t buflen);
/* name of the dominant kernel symbol of zp_ntt for this size (for matching rocprof output) */
int32_t zp_device_info_json(zp_ctx *ctx, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif
