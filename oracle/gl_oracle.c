/*
 * oracle/gl_oracle.c -- CPU restatement of the batch-prover hot path (TEST INFRASTRUCTURE).
 *
 * This file is the CHECKER, never the product: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path (eigen_zeth_amd/) never links
 * or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference (/root/reference, 0xEigenLabs/eigen-zeth) contains only the
 * gRPC *client* of the prover (src/prover/provider.rs:1-706, proto/prover/v1/prover.proto:1-195);
 * the arithmetic of the path lives in an external, un-pinned prover service.
 * There is no reference file, golden vector or known-answer test for any function below, so this
 * restatement follows the public algorithm definitions and is pinned by
 *   - mathematical identities (naive O(n^2) DFT with Python big-ints, direct polynomial
 *     evaluation, round trips) -- tests/test_oracle.py, tests/golden/
 *   - one external anchor: the Poseidon Grain-LFSR generator reproduces the published BN254 t=3
 *     first round constant, see tests/test_poseidon_constants.py.
 * Every unpinned choice (root of unity, coset shift, Poseidon tables, F_{p^3} modulus) is a
 * parameter here, exactly as in the product.
 *
 * Call sites in the reference that this path serves: src/prover/provider.rs:358-390
 * (GenChunkProof -> chunk STARK proofs), :422-451 (aggregation), :472-503 (final proof).
 *
 * Plain C11 + OpenMP.  Layout everywhere: column-major u64[W][N], canonical values < p.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef unsigned __int128 u128;

#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL /* 2^64 mod p */

/* ------------------------------------------------------------------ field */
static inline u64 gl_add(u64 a, u64 b) {
    u64 s = a + b;
    s += (0 - (u64)(s < a)) & GL_EPS; /* wrapped: +2^64 == +EPS (mod p), cannot wrap twice for a,b<p */
    u64 t = s - GL_P;
    return s >= GL_P ? t : s;         /* branch-free (cmov): operands are random, branches mispredict */
}
static inline u64 gl_sub(u64 a, u64 b) {
    u64 d = a - b;
    return d - ((0 - (u64)(a < b)) & GL_EPS); /* borrowed 2^64 == EPS too much */
}
static inline u64 gl_neg(u64 a) { return a ? GL_P - a : 0; }
static inline u64 gl_red128(u128 x) {
    u64 lo = (u64)x, hi = (u64)(x >> 64);
    u64 hh = hi >> 32, hl = hi & GL_EPS;
    /* x = lo + hl*2^64 + hh*2^96 == lo + hl*EPS - hh  (2^96 == -1) */
    u64 t0 = lo - hh;
    t0 -= (0 - (u64)(lo < hh)) & GL_EPS;
    u64 t1 = hl * GL_EPS;
    u64 r = t0 + t1;
    r += (0 - (u64)(r < t1)) & GL_EPS;
    u64 t = r - GL_P;
    return r >= GL_P ? t : r;
}
static inline u64 gl_mul(u64 a, u64 b) { return gl_red128((u128)a * b); }
static u64 gl_pow(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = gl_mul(r, b);
        b = gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
static inline u64 gl_inv(u64 a) { return gl_pow(a, GL_P - 2); }

u64 orc_add(u64 a, u64 b) { return gl_add(a, b); }
u64 orc_sub(u64 a, u64 b) { return gl_sub(a, b); }
u64 orc_mul(u64 a, u64 b) { return gl_mul(a, b); }
u64 orc_pow(u64 a, u64 e) { return gl_pow(a, e); }
u64 orc_inv(u64 a) { return gl_inv(a); }

/* root of the 2^logn subgroup derived from the configured 2^32-th root */
u64 orc_root(u64 root32, int logn) {
    u64 w = root32;
    for (int i = logn; i < 32; i++) w = gl_mul(w, w);
    return w;
}

/* ------------------------------------------------------------------ NTT (N1) */
static inline uint32_t bitrev32(uint32_t x, int bits) {
    uint32_t r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* in-place, natural in -> natural out:  X[k] = sum_n x[n] w^(nk)
 * tw: per-stage compact twiddles, tw[half + j] = w_(2*half)^j for j < half (half = 1,2,4,..,n/2) */
static void ntt_one(u64 *a, int logn, u64 w, const u64 *tw) {
    size_t n = (size_t)1 << logn;
    (void)w;
    for (size_t i = 0; i < n; i++) {
        size_t j = bitrev32((uint32_t)i, logn);
        if (i < j) { u64 t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int s = 1; s <= logn; s++) {
        size_t m = (size_t)1 << s, half = m >> 1;
        const u64 *ts = tw + half;
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                u64 t = gl_mul(ts[j], a[k + j + half]);
                u64 u = a[k + j];
                a[k + j] = gl_add(u, t);
                a[k + j + half] = gl_sub(u, t);
            }
    }
}

static u64 *make_tw(int logn, u64 w) {
    size_t n = (size_t)1 << logn;
    u64 *tw = (u64 *)malloc((n > 1 ? n : 2) * sizeof(u64));
    tw[0] = 0;
    tw[1] = 1;
    /* top stage directly, lower stages by striding the stage above: w_(m/2)^j = w_m^(2j) */
    if (logn >= 1) {
        size_t half = n >> 1;
        u64 c = 1;
        for (size_t j = 0; j < half; j++) { tw[half + j] = c; c = gl_mul(c, w); }
        for (size_t h = half >> 1; h >= 1; h >>= 1)
            for (size_t j = 0; j < h; j++) tw[h + j] = tw[2 * h + 2 * j];
    }
    return tw;
}

/* Cache-blocked form of the same transform for large N (what a tuned CPU prover does; the plain radix-2 loop above
 * walks the whole 128 MiB column 24 times): N = N1*N2, index i = i1*N2 + i2, output k = k1 + N1*k2,
 *   X[k1 + N1 k2] = sum_i2 w_N2^(i2 k2) w_N^(i2 k1) sum_i1 x[i1 N2 + i2] w_N1^(i1 k1).
 * Eight strided sub-vectors are gathered at a time so that every 64-byte line read is used whole; the sub-transforms
 * run on contiguous cache-resident buffers with ntt_one.  Exact arithmetic: bit-identical to ntt_one
 * (tests/test_oracle.py).  tmp: N elements, buf: 8*max(N1,N2) elements. */
static void ntt_one_blocked(u64 *a, int logn, u64 w, const u64 *tw1, const u64 *tw2, u64 *tmp, u64 *buf) {
    const int l1 = logn / 2, l2 = logn - l1;
    const size_t N1 = (size_t)1 << l1, N2 = (size_t)1 << l2;
    u64 wi2 = 1; /* w^i2 */
    for (size_t i2 = 0; i2 < N2; i2 += 8) {
        for (size_t i1 = 0; i1 < N1; i1++)
            for (int b = 0; b < 8; b++) buf[(size_t)b * N1 + i1] = a[i1 * N2 + i2 + b];
        for (int b = 0; b < 8; b++) {
            u64 *v = buf + (size_t)b * N1;
            ntt_one(v, l1, 0, tw1);
            u64 c = 1;
            for (size_t k1 = 0; k1 < N1; k1++) {
                tmp[(i2 + b) * N1 + k1] = gl_mul(v[k1], c);
                c = gl_mul(c, wi2);
            }
            wi2 = gl_mul(wi2, w);
        }
    }
    for (size_t k1 = 0; k1 < N1; k1 += 8) {
        for (size_t i2 = 0; i2 < N2; i2++)
            for (int b = 0; b < 8; b++) buf[(size_t)b * N2 + i2] = tmp[i2 * N1 + k1 + b];
        for (int b = 0; b < 8; b++) ntt_one(buf + (size_t)b * N2, l2, 0, tw2);
        for (size_t k2 = 0; k2 < N2; k2++)
            for (int b = 0; b < 8; b++) a[k1 + b + N1 * k2] = buf[(size_t)b * N2 + k2];
    }
}

#define ORC_BLOCKED_MIN_LOG 16
static int orc_force_simple = 0;
void orc_set_simple_ntt(int on) { orc_force_simple = on; }   /* tests: compare the two forms */

/* all W columns with root w (the caller passes the inverse root for the inverse transform) */
static void ntt_cols(u64 *cols, int logn, int W, u64 w) {
    const size_t n = (size_t)1 << logn;
    if (logn < ORC_BLOCKED_MIN_LOG || orc_force_simple) {
        u64 *tw = make_tw(logn, w);
#pragma omp parallel for schedule(dynamic, 1)
        for (int c = 0; c < W; c++) ntt_one(cols + (size_t)c * n, logn, w, tw);
        free(tw);
        return;
    }
    const int l1 = logn / 2, l2 = logn - l1;
    u64 *tw1 = make_tw(l1, gl_pow(w, (u64)1 << l2)), *tw2 = make_tw(l2, gl_pow(w, (u64)1 << l1));
#pragma omp parallel
    {
        u64 *tmp = (u64 *)malloc(n * sizeof(u64));
        u64 *buf = (u64 *)malloc(((size_t)8 << l2) * sizeof(u64));
#pragma omp for schedule(dynamic, 1)
        for (int c = 0; c < W; c++) ntt_one_blocked(cols + (size_t)c * n, logn, w, tw1, tw2, tmp, buf);
        free(tmp);
        free(buf);
    }
    free(tw1);
    free(tw2);
}

/* cols: u64[W][N] column-major, in place */
void orc_ntt(u64 *cols, int logn, int W, u64 root32) { ntt_cols(cols, logn, W, orc_root(root32, logn)); }

void orc_intt(u64 *cols, int logn, int W, u64 root32) {
    ntt_cols(cols, logn, W, gl_inv(orc_root(root32, logn)));
    size_t n = (size_t)1 << logn;
    u64 ninv = gl_inv((u64)n % GL_P);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < (size_t)W * n; i++) cols[i] = gl_mul(cols[i], ninv);
}

/* ------------------------------------------------------------------ LDE (N2)
 * in u64[W][N] evaluations on <w_N>;  out u64[W][bN] evaluations on shift*<w_bN>, natural order:
 *   c = iNTT_N(in);  c_i *= shift^i;  zero-pad to bN;  out = NTT_bN(c)                         */
void orc_lde(const u64 *in, u64 *out, int logn, int logb, int W, u64 shift, u64 root32) {
    size_t n = (size_t)1 << logn, m = (size_t)1 << (logn + logb);
    u64 wi = gl_inv(orc_root(root32, logn));
    u64 wm = orc_root(root32, logn + logb);
    u64 *twi = make_tw(logn, wi), *twm = make_tw(logn + logb, wm);
    u64 ninv = gl_inv((u64)n % GL_P);
    u64 *sp = (u64 *)malloc(n * sizeof(u64));
    u64 c = ninv;
    for (size_t i = 0; i < n; i++) { sp[i] = c; c = gl_mul(c, shift); }
#pragma omp parallel for schedule(dynamic, 1)
    for (int col = 0; col < W; col++) {
        u64 *o = out + (size_t)col * m;
        memcpy(o, in + (size_t)col * n, n * sizeof(u64));
        ntt_one(o, logn, wi, twi);
        for (size_t i = 0; i < n; i++) o[i] = gl_mul(o[i], sp[i]);
        memset(o + n, 0, (m - n) * sizeof(u64));
        ntt_one(o, logn + logb, wm, twm);
    }
    free(sp); free(twi); free(twm);
}

/* ------------------------------------------------------------------ Poseidon-12 (N3)
 * width 12 (rate 8 / capacity 4), S-box x^7, RF=8 (4+4) full and RP=22 partial rounds,
 * unoptimised textbook schedule: ARK -> S-box -> MDS per round.
 * rc: 30*12 round constants;  mds: 12x12 row-major, out[r] = sum_j mds[r][j]*in[j].        */
#define PW 12
#define PRF 8
#define PRP 22
static inline u64 sbox7(u64 x) {
    u64 x2 = gl_mul(x, x), x4 = gl_mul(x2, x2), x3 = gl_mul(x2, x);
    return gl_mul(x3, x4);
}
static void poseidon_perm(u64 st[PW], const u64 *rc, const u64 *mds) {
    for (int r = 0; r < PRF + PRP; r++) {
        for (int i = 0; i < PW; i++) st[i] = gl_add(st[i], rc[r * PW + i]);
        if (r < PRF / 2 || r >= PRF / 2 + PRP) {
            for (int i = 0; i < PW; i++) st[i] = sbox7(st[i]);
        } else {
            st[0] = sbox7(st[0]);
        }
        u64 nx[PW];
        for (int i = 0; i < PW; i++) {
            u64 acc = 0;
            for (int j = 0; j < PW; j++) acc = gl_add(acc, gl_mul(mds[i * PW + j], st[j]));
            nx[i] = acc;
        }
        memcpy(st, nx, sizeof(nx));
    }
}
void orc_poseidon_perm(u64 *states, size_t count, const u64 *rc, const u64 *mds) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; i++) poseidon_perm(states + i * PW, rc, mds);
}

/* Round-by-round trace of `count` permutations, the witness of a Poseidon AIR (32 rows per permutation):
 * states[e][32 k + r] = element e of the state BEFORE round r (r < 30), the output on rows 30 and 31;
 * cubes[e][32 k + r]  = (state + round constant)^3 on rows r < 30, state^3 on rows 30, 31.  Column stride = 32 * count. */
void orc_poseidon_trace(const u64 *inputs, size_t count, u64 *states, u64 *cubes, const u64 *rc, const u64 *mds) {
    const size_t n = 32 * count;
#pragma omp parallel for schedule(static)
    for (size_t k = 0; k < count; k++) {
        u64 st[PW];
        memcpy(st, inputs + k * PW, sizeof(st));
        for (int r = 0; r < 32; r++) {
            for (int i = 0; i < PW; i++) {
                const u64 x = r < PRF + PRP ? gl_add(st[i], rc[r * PW + i]) : st[i];
                states[(size_t)i * n + 32 * k + r] = st[i];
                cubes[(size_t)i * n + 32 * k + r] = gl_mul(gl_mul(x, x), x);
            }
            if (r >= PRF + PRP) continue;
            for (int i = 0; i < PW; i++) st[i] = gl_add(st[i], rc[r * PW + i]);
            if (r < PRF / 2 || r >= PRF / 2 + PRP) {
                for (int i = 0; i < PW; i++) st[i] = sbox7(st[i]);
            } else {
                st[0] = sbox7(st[0]);
            }
            u64 nx[PW];
            for (int i = 0; i < PW; i++) {
                u64 acc = 0;
                for (int j = 0; j < PW; j++) acc = gl_add(acc, gl_mul(mds[i * PW + j], st[j]));
                nx[i] = acc;
            }
            memcpy(st, nx, sizeof(nx));
        }
    }
}

/* linear hash of a row of `len` elements -> 4 elements.
 *  len <= 4 : identity, zero-padded (no permutation)
 *  else     : sponge, state = [rate 0..7 | capacity 8..11], capacity starts 0, each block of
 *             up to 8 elements OVERWRITES the rate (missing tail elements = 0), permute,
 *             the 4 outputs state[0..3] become the next block's capacity.                    */
static void linear_hash(const u64 *row, size_t len, u64 out[4], const u64 *rc, const u64 *mds) {
    if (len <= 4) {
        for (size_t i = 0; i < 4; i++) out[i] = i < len ? row[i] : 0;
        return;
    }
    u64 st[PW];
    u64 cap[4] = {0, 0, 0, 0};
    for (size_t off = 0; off < len; off += 8) {
        for (int i = 0; i < 8; i++) st[i] = (off + i < len) ? row[off + i] : 0;
        for (int i = 0; i < 4; i++) st[8 + i] = cap[i];
        poseidon_perm(st, rc, mds);
        for (int i = 0; i < 4; i++) cap[i] = st[i];
    }
    for (int i = 0; i < 4; i++) out[i] = cap[i];
}
static void hash_pair(const u64 *l, const u64 *r, u64 out[4], const u64 *rc, const u64 *mds) {
    u64 st[PW];
    for (int i = 0; i < 4; i++) { st[i] = l[i]; st[4 + i] = r[i]; st[8 + i] = 0; }
    poseidon_perm(st, rc, mds);
    for (int i = 0; i < 4; i++) out[i] = st[i];
}

/* Merkle commitment over M rows of a column-major matrix u64[W][M] (leaf i = linear hash of
 * row i across the W columns).  tree: (2M-1)*4 u64, level 0 (M leaves) first, then M/2 ... 1;
 * the root is the last 4 elements.  M must be a power of two.                                */
void orc_merkle_commit(const u64 *cols, size_t M, int W, u64 *tree, const u64 *rc, const u64 *mds) {
#pragma omp parallel
    {
        u64 *row = (u64 *)malloc((size_t)(W > 0 ? W : 1) * sizeof(u64));
#pragma omp for schedule(static)
        for (size_t i = 0; i < M; i++) {
            for (int c = 0; c < W; c++) row[c] = cols[(size_t)c * M + i];
            linear_hash(row, (size_t)W, tree + i * 4, rc, mds);
        }
        free(row);
    }
    u64 *prev = tree;
    size_t cnt = M;
    while (cnt > 1) {
        u64 *next = prev + cnt * 4;
        size_t half = cnt >> 1;
#pragma omp parallel for schedule(static) if (half > 64)
        for (size_t i = 0; i < half; i++)
            hash_pair(prev + (2 * i) * 4, prev + (2 * i + 1) * 4, next + i * 4, rc, mds);
        prev = next;
        cnt = half;
    }
}

/* leaves already given as rows of `len` contiguous elements (row-major [M][len]) */
void orc_merkle_commit_rows(const u64 *rows, size_t M, size_t len, u64 *tree, const u64 *rc,
                            const u64 *mds) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < M; i++) linear_hash(rows + i * len, len, tree + i * 4, rc, mds);
    u64 *prev = tree;
    size_t cnt = M;
    while (cnt > 1) {
        u64 *next = prev + cnt * 4;
        size_t half = cnt >> 1;
#pragma omp parallel for schedule(static) if (half > 64)
        for (size_t i = 0; i < half; i++)
            hash_pair(prev + (2 * i) * 4, prev + (2 * i + 1) * 4, next + i * 4, rc, mds);
        prev = next;
        cnt = half;
    }
}

void orc_linear_hash(const u64 *row, size_t len, u64 *out4, const u64 *rc, const u64 *mds) {
    linear_hash(row, len, out4, rc, mds);
}

/* authentication path of leaf idx: log2(M) siblings of 4 elements, bottom-up */
void orc_merkle_path(const u64 *tree, size_t M, size_t idx, u64 *path) {
    const u64 *lvl = tree;
    size_t cnt = M;
    int d = 0;
    while (cnt > 1) {
        memcpy(path + 4 * d, lvl + (idx ^ 1) * 4, 4 * sizeof(u64));
        lvl += cnt * 4;
        cnt >>= 1;
        idx >>= 1;
        d++;
    }
}

int orc_merkle_verify(const u64 *leaf4, size_t M, size_t idx, const u64 *path, const u64 *root,
                      const u64 *rc, const u64 *mds) {
    u64 cur[4];
    memcpy(cur, leaf4, sizeof(cur));
    int d = 0;
    for (size_t cnt = M; cnt > 1; cnt >>= 1, idx >>= 1, d++) {
        u64 nx[4];
        if (idx & 1) hash_pair(path + 4 * d, cur, nx, rc, mds);
        else hash_pair(cur, path + 4 * d, nx, rc, mds);
        memcpy(cur, nx, sizeof(cur));
    }
    return memcmp(cur, root, sizeof(cur)) == 0;
}

/* ------------------------------------------------------------------ F_{p^3} = F_p[x]/(x^3 - x - 1) */
typedef struct { u64 c[3]; } e3;
static inline e3 e3_add(e3 a, e3 b) { e3 r = {{gl_add(a.c[0], b.c[0]), gl_add(a.c[1], b.c[1]), gl_add(a.c[2], b.c[2])}}; return r; }
static inline e3 e3_sub(e3 a, e3 b) { e3 r = {{gl_sub(a.c[0], b.c[0]), gl_sub(a.c[1], b.c[1]), gl_sub(a.c[2], b.c[2])}}; return r; }
static inline e3 e3_mul(e3 a, e3 b) {
    /* schoolbook, then x^3 = x + 1, x^4 = x^2 + x */
    u64 d0 = gl_mul(a.c[0], b.c[0]);
    u64 d1 = gl_add(gl_mul(a.c[0], b.c[1]), gl_mul(a.c[1], b.c[0]));
    u64 d2 = gl_add(gl_add(gl_mul(a.c[0], b.c[2]), gl_mul(a.c[1], b.c[1])), gl_mul(a.c[2], b.c[0]));
    u64 d3 = gl_add(gl_mul(a.c[1], b.c[2]), gl_mul(a.c[2], b.c[1]));
    u64 d4 = gl_mul(a.c[2], b.c[2]);
    e3 r = {{gl_add(d0, d3), gl_add(gl_add(d1, d3), d4), gl_add(d2, d4)}};
    return r;
}
static inline e3 e3_scale(e3 a, u64 s) { e3 r = {{gl_mul(a.c[0], s), gl_mul(a.c[1], s), gl_mul(a.c[2], s)}}; return r; }
void orc_e3_mul(const u64 *a, const u64 *b, u64 *out) {
    e3 x = {{a[0], a[1], a[2]}}, y = {{b[0], b[1], b[2]}};
    e3 r = e3_mul(x, y);
    out[0] = r.c[0]; out[1] = r.c[1]; out[2] = r.c[2];
}
static e3 e3_pow(e3 b, const u64 e[3]) { /* exponent up to 192 bits, little-endian limbs */
    e3 r = {{1, 0, 0}};
    for (int l = 2; l >= 0; l--)
        for (int i = 63; i >= 0; i--) {
            r = e3_mul(r, r);
            if ((e[l] >> i) & 1) r = e3_mul(r, b);
        }
    return r;
}
/* inverse via a^(p^3-2); exponent p^3-2 is passed by the caller (python big-int -> 3 limbs) */
void orc_e3_pow(const u64 *a, const u64 *e3limbs, u64 *out) {
    e3 x = {{a[0], a[1], a[2]}};
    e3 r = e3_pow(x, e3limbs);
    out[0] = r.c[0]; out[1] = r.c[1]; out[2] = r.c[2];
}

/* ------------------------------------------------------------------ FRI fold (N5)
 * layer_in : evaluations of a polynomial f (values in F_{p^3}, stored as 3 planes u64[3][n],
 *            plane-major) on the coset  shift*<w_n>, natural order.
 * fold by 2^logf:  f(x) = sum_{j<2^logf} x^j g_j(x^(2^logf));  out(y) = sum_j beta^j g_j(y)
 *            evaluated on (shift^(2^logf))*<w_{n/2^logf}>, natural order, planes u64[3][n>>logf].
 * Computation per output index i (m = n>>logf): the 2^logf values f(shift*w_n^(i + m*k)),
 * k<2^logf, are the evaluations of the degree<2^logf polynomial  h_i(z)=sum_j g_j(y_i) z^j  on
 * the coset (shift*w_n^i)*<w_{2^logf}>;  interpolate (iNTT + coset unscale) then Horner at beta. */
void orc_fri_fold(const u64 *in, u64 *out, int logn, int logf, const u64 *beta3, u64 shift,
                  u64 root32) {
    size_t n = (size_t)1 << logn, f = (size_t)1 << logf, m = n >> logf;
    u64 wn = orc_root(root32, logn);
    u64 wf_inv = gl_inv(orc_root(root32, logf));
    u64 finv = gl_inv((u64)f);
    e3 beta = {{beta3[0], beta3[1], beta3[2]}};
    u64 wfp[64];
    wfp[0] = 1;
    for (size_t k = 1; k < f; k++) wfp[k] = gl_mul(wfp[k - 1], wf_inv);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) {
        e3 v[64], cof[64];
        for (size_t k = 0; k < f; k++)
            for (int c = 0; c < 3; c++) v[k].c[c] = in[(size_t)c * n + i + m * k];
        /* naive inverse DFT of size f (f <= 64): cof_j = 1/f * sum_k v_k wf^(-jk) */
        for (size_t j = 0; j < f; j++) {
            e3 acc = {{0, 0, 0}};
            for (size_t k = 0; k < f; k++) acc = e3_add(acc, e3_scale(v[k], wfp[(j * k) % f]));
            cof[j] = e3_scale(acc, finv);
        }
        /* unscale the coset: h_i coefficient j = cof_j / (shift*w_n^i)^j */
        u64 xi_inv = gl_inv(gl_mul(shift, gl_pow(wn, i)));
        u64 sc = 1;
        e3 acc = {{0, 0, 0}}, bp = {{1, 0, 0}};
        for (size_t j = 0; j < f; j++) {
            acc = e3_add(acc, e3_mul(e3_scale(cof[j], sc), bp));
            sc = gl_mul(sc, xi_inv);
            bp = e3_mul(bp, beta);
        }
        for (int c = 0; c < 3; c++) out[(size_t)c * m + i] = acc.c[c];
    }
}

/* ------------------------------------------------------------------ DEEP quotient (definition level)
 * F(x) = sum_{k<Wa+Wb} g^k (p_k(x)-e_k)/(x-z) + sum_{k<n_next} g^(Wa+Wb+k) (p_k(x)-e'_k)/(x-zw),
 * x = shift*w_M^r.  Every term is divided separately (inverse by a^(p^3-2)).                     */
static e3 e3_inv_pow(e3 a) {
    /* p^3 - 2, little-endian 64-bit limbs */
    static const u64 E[3] = {0xfffffffcffffffffULL, 0xfffffff900000005ULL, 0xfffffffd00000005ULL};
    return e3_pow(a, E);
}
void orc_e3_inv(const u64 *a, u64 *out) {
    e3 x = {{a[0], a[1], a[2]}};
    e3 r = e3_inv_pow(x);
    out[0] = r.c[0]; out[1] = r.c[1]; out[2] = r.c[2];
}
void orc_deep_quotient(const u64 *cols_a, int Wa, const u64 *cols_b, int Wb, int logm, int n_next,
                       const u64 *z, const u64 *zw, const u64 *gamma, const u64 *ev_z, const u64 *ev_zw,
                       u64 shift, u64 root32, u64 *out) {
    size_t M = (size_t)1 << logm;
    int W = Wa + Wb;
    u64 wm = orc_root(root32, logm);
    e3 g = {{gamma[0], gamma[1], gamma[2]}};
    e3 *gp = (e3 *)malloc((size_t)(W + n_next) * sizeof(e3));
    e3 cur = {{1, 0, 0}};
    for (int k = 0; k < W + n_next; k++) { gp[k] = cur; cur = e3_mul(cur, g); }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < M; r++) {
        u64 x = gl_mul(shift, gl_pow(wm, r));
        e3 d1 = {{gl_sub(x, z[0]), gl_neg(z[1]), gl_neg(z[2])}};
        e3 d2 = {{gl_sub(x, zw[0]), gl_neg(zw[1]), gl_neg(zw[2])}};
        e3 i1 = e3_inv_pow(d1), i2 = e3_inv_pow(d2);
        e3 acc = {{0, 0, 0}};
        for (int k = 0; k < W; k++) {
            u64 v = k < Wa ? cols_a[(size_t)k * M + r] : cols_b[(size_t)(k - Wa) * M + r];
            e3 num = {{gl_sub(v, ev_z[k * 3]), gl_neg(ev_z[k * 3 + 1]), gl_neg(ev_z[k * 3 + 2])}};
            acc = e3_add(acc, e3_mul(gp[k], e3_mul(num, i1)));
            if (k < n_next) {
                e3 n2 = {{gl_sub(v, ev_zw[k * 3]), gl_neg(ev_zw[k * 3 + 1]), gl_neg(ev_zw[k * 3 + 2])}};
                acc = e3_add(acc, e3_mul(gp[W + k], e3_mul(n2, i2)));
            }
        }
        for (int c = 0; c < 3; c++) out[(size_t)c * M + r] = acc.c[c];
    }
    free(gp);
}

/* ------------------------------------------------------------------ polynomial helpers used by tests */
/* evaluate polynomial with base-field coefficients (ascending) at base point x */
u64 orc_poly_eval(const u64 *coef, size_t n, u64 x) {
    u64 acc = 0;
    for (size_t i = n; i-- > 0;) acc = gl_add(gl_mul(acc, x), coef[i]);
    return acc;
}
/* evaluate polynomial with base-field coefficients at an F_{p^3} point */
void orc_poly_eval_e3(const u64 *coef, size_t n, const u64 *x3, u64 *out3) {
    e3 x = {{x3[0], x3[1], x3[2]}}, acc = {{0, 0, 0}};
    for (size_t i = n; i-- > 0;) {
        acc = e3_mul(acc, x);
        acc.c[0] = gl_add(acc.c[0], coef[i]);
    }
    out3[0] = acc.c[0]; out3[1] = acc.c[1]; out3[2] = acc.c[2];
}

/* W polynomials (base coefficients u64[W][n]) at an F_{p^3} point: out[W][3] (Horner) */
void orc_poly_eval_e3_cols(const u64 *coef, size_t n, int W, const u64 *x3, u64 *out) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int c = 0; c < W; c++) orc_poly_eval_e3(coef + (size_t)c * n, n, x3, out + 3 * c);
}

/* same DEEP quotient with the adjugate inverse (fast path for the CPU baseline; the pow-based
 * orc_deep_quotient above stays the definition it is tested against) */
static e3 e3_inv_adj(e3 a) {
    u64 a0 = a.c[0], a1 = a.c[1], a2 = a.c[2];
    u64 s02 = gl_add(a0, a2), s12 = gl_add(a1, a2);
    u64 c00 = gl_sub(gl_mul(s02, s02), gl_mul(s12, a1));
    u64 c01 = gl_sub(gl_mul(s12, a2), gl_mul(a1, s02));
    u64 c02 = gl_sub(gl_mul(a1, a1), gl_mul(s02, a2));
    u64 det = gl_add(gl_add(gl_mul(a0, c00), gl_mul(a2, c01)), gl_mul(a1, c02));
    u64 di = gl_inv(det);
    e3 r = {{gl_mul(c00, di), gl_mul(c01, di), gl_mul(c02, di)}};
    return r;
}
void orc_deep_quotient_fast(const u64 *cols_a, int Wa, const u64 *cols_b, int Wb, int logm, int n_next,
                            const u64 *z, const u64 *zw, const u64 *gamma, const u64 *ev_z, const u64 *ev_zw,
                            u64 shift, u64 root32, u64 *out) {
    size_t M = (size_t)1 << logm;
    int W = Wa + Wb;
    u64 wm = orc_root(root32, logm);
    e3 g = {{gamma[0], gamma[1], gamma[2]}};
    e3 *gp = (e3 *)malloc((size_t)(W + n_next) * sizeof(e3));
    e3 cur = {{1, 0, 0}}, ca = {{0, 0, 0}}, cb = {{0, 0, 0}};
    for (int k = 0; k < W + n_next; k++) {
        gp[k] = cur;
        if (k < W) { e3 e = {{ev_z[3 * k], ev_z[3 * k + 1], ev_z[3 * k + 2]}}; ca = e3_add(ca, e3_mul(cur, e)); }
        else { int j = k - W; e3 e = {{ev_zw[3 * j], ev_zw[3 * j + 1], ev_zw[3 * j + 2]}}; cb = e3_add(cb, e3_mul(cur, e)); }
        cur = e3_mul(cur, g);
    }
#pragma omp parallel
    {
#pragma omp for schedule(static)
        for (size_t r = 0; r < M; r++) {
            u64 x = gl_mul(shift, gl_pow(wm, r));
            e3 A = {{0, 0, 0}}, B = {{0, 0, 0}};
            for (int k = 0; k < W; k++) {
                u64 v = k < Wa ? cols_a[(size_t)k * M + r] : cols_b[(size_t)(k - Wa) * M + r];
                A = e3_add(A, e3_scale(gp[k], v));
                if (k < n_next) B = e3_add(B, e3_scale(gp[W + k], v));
            }
            A = e3_sub(A, ca);
            B = e3_sub(B, cb);
            e3 d1 = {{gl_sub(x, z[0]), gl_neg(z[1]), gl_neg(z[2])}};
            e3 Fv = e3_mul(A, e3_inv_adj(d1));
            if (n_next > 0) {
                e3 d2 = {{gl_sub(x, zw[0]), gl_neg(zw[1]), gl_neg(zw[2])}};
                Fv = e3_add(Fv, e3_mul(B, e3_inv_adj(d2)));
            }
            for (int c = 0; c < 3; c++) out[(size_t)c * M + r] = Fv.c[c];
        }
    }
    free(gp);
}

/* grand product column (sequential definition): Z[0]=1, Z[i+1] = Z[i]*(a[i]+g)/(b[i]+g) in F_{p^3} */
void orc_grand_product(const u64 *a, const u64 *b, size_t n, const u64 *g, u64 *out) {
    e3 z = {{1, 0, 0}};
    for (size_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) out[(size_t)c * n + i] = z.c[c];
        e3 num = {{gl_add(a[i], g[0]), g[1], g[2]}}, den = {{gl_add(b[i], g[0]), g[1], g[2]}};
        z = e3_mul(z, e3_mul(num, e3_inv_pow(den)));
    }
}

/* LogUp lookup columns (sequential definition): h1 = 1/(a+g), h2 = m/(t+g), S[0] = 0, S[i+1] = S[i] + h1[i] - h2[i];
 * out u64[9][n]: planes h1 0..2, h2 3..5, S 6..8.  Inversion by Fermat (e3_inv_pow), independent of the GPU's adjugate form. */
void orc_logup_columns(const u64 *a, const u64 *t, const u64 *m, size_t n, const u64 *g, u64 *out) {
    e3 s = {{0, 0, 0}};
    for (size_t i = 0; i < n; i++) {
        e3 da = {{gl_add(a[i], g[0]), g[1], g[2]}}, dt = {{gl_add(t[i], g[0]), g[1], g[2]}};
        e3 h1 = e3_inv_pow(da), h2 = e3_inv_pow(dt);
        for (int c = 0; c < 3; c++) h2.c[c] = gl_mul(h2.c[c], m[i]);
        for (int c = 0; c < 3; c++) {
            out[(size_t)c * n + i] = h1.c[c];
            out[(size_t)(3 + c) * n + i] = h2.c[c];
            out[(size_t)(6 + c) * n + i] = s.c[c];
            s.c[c] = gl_add(s.c[c], gl_sub(h1.c[c], h2.c[c]));
        }
    }
}

/* ---------------------------------------------------------------- constraint program interpreter (N4)
 * The AIR arrives as data: the "constraint program" blob of include/zeth_prover.h (header of 12 words, constants,
 * three-address instructions over a small slot file, OUT marks constraint k, stage-2 table).  This interpreter is
 * written against that layout only -- it shares nothing with the product's code generator or GPU interpreter.
 * Per LDE row r: x = shift * wM^r, xml = x - wlast, rn = (r + b) mod M;  out[c][r] = (sum_k alpha^k C_k) * zhinv[r mod b].
 * returns 0, or -1 for a malformed program. */
int orc_quotient_program_rows(const u64 *prog, size_t prog_len, const u64 *cols, size_t sc, const u64 *fixedc, size_t sf, size_t M,
                              size_t b, size_t row0, size_t nrows, const u64 *pub, const u64 *apow, const u64 *zhinv, u64 shift, u64 wM,
                              u64 wlast, u64 *out, size_t so);
int orc_quotient_program(const u64 *prog, size_t prog_len, const u64 *cols, const u64 *fixedc, size_t M, size_t b,
                         const u64 *pub, const u64 *apow, const u64 *zhinv, u64 shift, u64 wM, u64 wlast, u64 *out) {
    return orc_quotient_program_rows(prog, prog_len, cols, M, fixedc, M, M, b, 0, M, pub, apow, zhinv, shift, wM, wlast, out, M);
}
/* row window [row0, row0 + nrows) of the M-row domain (one row shard): local row j = domain row row0 + j, columns with
 * strides sc / sf / so; unless the window is the whole domain the b halo rows follow each column (next row = j + b) */
int orc_quotient_program_rows(const u64 *prog, size_t prog_len, const u64 *cols, size_t sc, const u64 *fixedc, size_t sf, size_t M,
                              size_t b, size_t row0, size_t nrows, const u64 *pub, const u64 *apow, const u64 *zhinv, u64 shift, u64 wM,
                              u64 wlast, u64 *out, size_t so) {
    const int whole = row0 == 0 && nrows == M;
    if (prog_len < 12) return -1;
    static const unsigned char magic[8] = {'Z', 'P', 'A', 'I', 'R', '1', 0, 0};
    if (memcmp(prog, magic, 8) != 0) return -1;
    const size_t n_const = prog[6], n_instr = prog[7], n_slots = prog[9], n_s2 = prog[10];
    if (n_slots == 0 || n_slots > 4096 || prog[3] < 2) return -1;
    {   /* behind the stage-2 table: the sparse periodic fixed columns (2..n_fixed-1), [lp | n << 8] + n (pos, value) pairs each.
           This evaluator takes them MATERIALISED (fixedc holds n_fixed full columns); only the length is checked here. */
        size_t at = 12 + n_const + n_instr + 4 * n_s2;
        for (u64 k = 2; k < prog[3]; k++) {
            if (at >= prog_len) return -1;
            at += 1 + 2 * (size_t)(prog[at] >> 8);
        }
        if (at != prog_len) return -1;
    }
    const u64 *consts = prog + 12, *ins = consts + n_const;
    int bad = 0;
#pragma omp parallel
    {
        u64 *slots = (u64 *)malloc(n_slots * sizeof(u64));
#pragma omp for schedule(static)
        for (size_t r = 0; r < nrows; r++) {
            const size_t rn = whole ? ((r + b) & (M - 1)) : r + b;
            const u64 x = gl_mul(shift, gl_pow(wM, (u64)(row0 + r)));
            const u64 xml = gl_sub(x, wlast);
            u64 acc[3] = {0, 0, 0};
            size_t k_out = 0;
            for (size_t i = 0; i < n_instr; i++) {
                const u64 w = ins[i];
                const unsigned op = (unsigned)(w & 0xFF), dst = (unsigned)((w >> 8) & 0xFFFF);
                u64 v[2];
                for (int o = 0; o < 2; o++) {
                    const unsigned kind = (unsigned)((w >> (24 + 20 * o)) & 0xF), idx = (unsigned)((w >> (28 + 20 * o)) & 0xFFFF);
                    switch (kind) {
                        case 0: v[o] = slots[idx]; break;
                        case 1: v[o] = cols[(size_t)idx * sc + r]; break;
                        case 2: v[o] = cols[(size_t)idx * sc + rn]; break;
                        case 3: v[o] = fixedc[(size_t)idx * sf + r]; break;
                        case 4: v[o] = pub[idx]; break;
                        case 5: v[o] = n_const ? consts[idx] : 0; break;
                        case 6: v[o] = xml; break;
                        default: v[o] = 0; bad = 1;
                    }
                    if (op == 4) break;   /* OUT has one operand */
                }
                if (op == 1) slots[dst] = gl_add(v[0], v[1]);
                else if (op == 2) slots[dst] = gl_sub(v[0], v[1]);
                else if (op == 3) slots[dst] = gl_mul(v[0], v[1]);
                else if (op == 4) {
                    for (int c = 0; c < 3; c++) acc[c] = gl_add(acc[c], gl_mul(v[0], apow[3 * k_out + c]));
                    k_out++;
                } else bad = 1;
            }
            const u64 zi = zhinv[r & (b - 1)];
            for (int c = 0; c < 3; c++) out[(size_t)c * so + r] = gl_mul(acc[c], zi);
        }
        free(slots);
    }
    return bad ? -1 : 0;
}

/* proof-of-work grinding before the query phase: the smallest nonce n with
 * Poseidon(seed[0..3] || n || 0^7)[0] >> (64 - bits) == 0.  Blocks of 4096 nonces are searched in parallel, in order. */
void orc_poseidon_perm(u64 *states, size_t count, const u64 *rc, const u64 *mds);
u64 orc_pow_grind(const u64 *seed4, int bits, const u64 *rc, const u64 *mds) {
    if (bits <= 0) return 0;
    enum { BLK = 4096 };
    u64 *st = (u64 *)malloc((size_t)BLK * 12 * sizeof(u64));
    for (u64 base = 0;; base += BLK) {
        for (size_t i = 0; i < BLK; i++) {
            u64 *s = st + 12 * i;
            memcpy(s, seed4, 4 * sizeof(u64));
            s[4] = base + i;
            memset(s + 5, 0, 7 * sizeof(u64));
        }
        orc_poseidon_perm(st, BLK, rc, mds);
        for (size_t i = 0; i < BLK; i++)
            if ((st[12 * i] >> (64 - bits)) == 0) {
                free(st);
                return base + i;
            }
    }
}

/* coefficients of p(shift * X): cols[c][i] *= shift^i  (what the LDE has between its two transforms) */
void orc_coset_scale(u64 *cols, size_t n, int W, u64 shift) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < W; c++) {
        u64 s = 1;
        for (size_t i = 0; i < n; i++) {
            cols[(size_t)c * n + i] = gl_mul(cols[(size_t)c * n + i], s);
            s = gl_mul(s, shift);
        }
    }
}

/* DEEP quotient on a row window [row0, row0 + nrows) of the 2^logm-row domain (one row shard), strides in elements */
void orc_deep_quotient_rows(const u64 *cols_a, int Wa, size_t sa, const u64 *cols_b, int Wb, size_t sb, int logm, size_t row0,
                            size_t nrows, int n_next, const u64 *z, const u64 *zw, const u64 *gamma, const u64 *ev_z,
                            const u64 *ev_zw, u64 shift, u64 root32, u64 *out, size_t so) {
    int W = Wa + Wb;
    u64 wm = orc_root(root32, logm);
    e3 g = {{gamma[0], gamma[1], gamma[2]}};
    e3 *gp = (e3 *)malloc((size_t)(W + n_next) * sizeof(e3));
    e3 cur = {{1, 0, 0}};
    for (int k = 0; k < W + n_next; k++) { gp[k] = cur; cur = e3_mul(cur, g); }
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < nrows; r++) {
        u64 x = gl_mul(shift, gl_pow(wm, (u64)(row0 + r)));
        e3 d1 = {{gl_sub(x, z[0]), gl_neg(z[1]), gl_neg(z[2])}};
        e3 d2 = {{gl_sub(x, zw[0]), gl_neg(zw[1]), gl_neg(zw[2])}};
        e3 i1 = e3_inv_pow(d1), i2 = e3_inv_pow(d2);
        e3 acc = {{0, 0, 0}};
        for (int k = 0; k < W; k++) {
            u64 v = k < Wa ? cols_a[(size_t)k * sa + r] : cols_b[(size_t)(k - Wa) * sb + r];
            e3 num = {{gl_sub(v, ev_z[k * 3]), gl_neg(ev_z[k * 3 + 1]), gl_neg(ev_z[k * 3 + 2])}};
            acc = e3_add(acc, e3_mul(gp[k], e3_mul(num, i1)));
            if (k < n_next) {
                e3 n2 = {{gl_sub(v, ev_zw[k * 3]), gl_neg(ev_zw[k * 3 + 1]), gl_neg(ev_zw[k * 3 + 2])}};
                acc = e3_add(acc, e3_mul(gp[W + k], e3_mul(n2, i2)));
            }
        }
        for (int c = 0; c < 3; c++) out[(size_t)c * so + r] = acc.c[c];
    }
    free(gp);
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
