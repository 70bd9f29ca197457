/* oracle/bn254_hash.c -- CPU restatement of the BN128-hash mode (TEST INFRASTRUCTURE, see gl_oracle.c header):
 * Poseidon over the BN254 scalar field (x^5, 8 full + rp partial rounds, TEXTBOOK dense schedule) and the 16-ary
 * Merkle tree over Goldilocks rows packed three to a field element.  Restates oracle/naive.py (poseidon_bn254_perm,
 * merkle16_leaf, merkle16_tree) with 4 x 64-bit Montgomery arithmetic so that whole BN128-mode STARK proofs can be
 * produced and checked on the CPU in seconds; pinned to the big-int definitions by tests/test_oracle.py.
 * No reference counterpart; PARITY UNPINNED with respect to the external prover.
 * Field elements cross this interface as 4 little-endian u64 words, standard form, < r. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 RMOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 RINV = 0xc2e1f593efffffffULL;      /* -r^-1 mod 2^64 */
/* 2^512 mod r */
static const u64 RR[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};

typedef struct { u64 w[4]; } fe;

static int ge_r(const u64 *a) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] != RMOD[i]) return a[i] > RMOD[i];
    }
    return 1;
}
static void sub_r(u64 *a) {
    u128 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a[i] - RMOD[i] - br;
        a[i] = (u64)d;
        br = (d >> 64) & 1;
    }
}
static fe fe_add(fe a, fe b) {
    fe r;
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.w[i] + b.w[i];
        r.w[i] = (u64)c;
        c >>= 64;
    }
    if (c || ge_r(r.w)) sub_r(r.w);   /* a, b < r < 2^254: no carry out in fact */
    return r;
}
/* Montgomery product a b / 2^256 mod r (CIOS) */
static fe fe_mul(fe a, fe b) {
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.w[j] * b.w[i] + t[j];
            t[j] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (u64)c;
        t[5] = (u64)(c >> 64);
        const u64 m = t[0] * RINV;
        c = (u128)m * RMOD[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * RMOD[j] + t[j];
            t[j - 1] = (u64)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (u64)c;
        t[4] = t[5] + (u64)(c >> 64);
    }
    fe r;
    memcpy(r.w, t, 32);
    if (t[4] || ge_r(r.w)) sub_r(r.w);
    return r;
}
static fe fe_from_words(const u64 *w) { fe a, rr; memcpy(a.w, w, 32); memcpy(rr.w, RR, 32); return fe_mul(a, rr); }
static void fe_to_words(fe a, u64 *w) { fe one = {{1, 0, 0, 0}}; fe s = fe_mul(a, one); memcpy(w, s.w, 32); }
static fe fe_pow5(fe x) { fe x2 = fe_mul(x, x), x4 = fe_mul(x2, x2); return fe_mul(x4, x); }

#define MAXT 17
typedef struct { int t, rp; fe *rc; fe mds[MAXT * MAXT]; } p254_table;
static p254_table TAB[2];     /* [0]: t = 3, [1]: t = 17 */

/* install the tables of one width: rc (8 + rp) * t elements round-major, mds t * t row-major (4 words each, < r) */
int orc_p254_set(int t, int rp, const u64 *rc, const u64 *mds) {
    if ((t != 3 && t != 17) || rp < 1 || rp > 128) return -1;
    p254_table *tb = &TAB[t == 3 ? 0 : 1];
    free(tb->rc);
    tb->rc = (fe *)malloc(sizeof(fe) * (size_t)(8 + rp) * t);
    if (!tb->rc) return -2;
    for (int i = 0; i < (8 + rp) * t; i++) { if (ge_r(rc + 4 * i)) return -3; tb->rc[i] = fe_from_words(rc + 4 * i); }
    for (int i = 0; i < t * t; i++) { if (ge_r(mds + 4 * i)) return -3; tb->mds[i] = fe_from_words(mds + 4 * i); }
    tb->t = t;
    tb->rp = rp;
    return 0;
}

/* textbook schedule (naive.py: poseidon_bn254_perm): ARK -> x^5 (all / element 0) -> dense matrix */
static void perm(const p254_table *tb, fe *s) {
    const int t = tb->t;
    fe n[MAXT];
    for (int r = 0; r < 8 + tb->rp; r++) {
        for (int i = 0; i < t; i++) s[i] = fe_add(s[i], tb->rc[r * t + i]);
        if (r < 4 || r >= 4 + tb->rp) {
            for (int i = 0; i < t; i++) s[i] = fe_pow5(s[i]);
        } else {
            s[0] = fe_pow5(s[0]);
        }
        for (int i = 0; i < t; i++) {
            fe acc = {{0, 0, 0, 0}};
            for (int j = 0; j < t; j++) acc = fe_add(acc, fe_mul(tb->mds[i * t + j], s[j]));
            n[i] = acc;
        }
        memcpy(s, n, sizeof(fe) * t);
    }
}

/* states: count x t elements (4 words each), permuted in place */
int orc_p254_perm(u64 *states, size_t count, int t) {
    const p254_table *tb = &TAB[t == 3 ? 0 : 1];
    if ((t != 3 && t != 17) || tb->t != t) return -1;
#pragma omp parallel for schedule(static)
    for (size_t c = 0; c < count; c++) {
        fe s[MAXT];
        for (int i = 0; i < t; i++) s[i] = fe_from_words(states + (c * t + i) * 4);
        perm(tb, s);
        for (int i = 0; i < t; i++) fe_to_words(s[i], states + (c * t + i) * 4);
    }
    return 0;
}

/* leaf of row `row` (len Goldilocks values, stride between consecutive values): naive.py merkle16_leaf */
static fe leaf_of(const p254_table *tb, const u64 *row, size_t len, size_t stride) {
    fe cap = {{0, 0, 0, 0}};
    for (size_t base = 0; base < len || base == 0; base += 56) {      /* 56 values per permutation: naive.py pack_leaf_block */
        fe s[MAXT];
        s[0] = cap;
        for (size_t e = 0; e < 16; e++) {
            u64 w[4] = {0, 0, 0, 0};
            for (int c = 0; c < 3; c++)
                if (base + 3 * e + c < len) w[c] = row[(base + 3 * e + c) * stride];
            const size_t x = base + 48 + (e >> 1);
            if (x < len) w[3] = (e & 1) ? (row[x * stride] >> 32) : (row[x * stride] & 0xFFFFFFFFULL);
            s[1 + e] = fe_from_words(w);      /* < 2^224 < r */
        }
        perm(tb, s);
        cap = s[0];
    }
    return cap;
}

static size_t nodes16(size_t M) { size_t n = M, tot = M; while (n > 1) { n = (n + 15) / 16; tot += n; } return tot; }
size_t orc_merkle16_nodes(size_t M) { return nodes16(M); }

/* cols u64[W][M] column-major; tree u64[nodes][4]: leaves, then each level, root last (naive.py merkle16_tree) */
int orc_merkle16_tree(const u64 *cols, size_t M, int W, u64 *tree) {
    const p254_table *tb = &TAB[1];
    if (tb->t != 17 || M < 1 || W < 1) return -1;
    fe *dig = (fe *)malloc(sizeof(fe) * nodes16(M));
    if (!dig) return -2;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < M; i++) dig[i] = leaf_of(tb, cols + i, (size_t)W, M);
    size_t n = M, off = 0;
    while (n > 1) {
        const size_t nn = (n + 15) / 16;
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < nn; i++) {
            fe s[MAXT];
            memset(s, 0, sizeof s);
            for (size_t c = 0; c < 16 && i * 16 + c < n; c++) s[1 + c] = dig[off + i * 16 + c];
            perm(tb, s);
            dig[off + n + i] = s[0];
        }
        off += n;
        n = nn;
    }
    const size_t tot = nodes16(M);
    for (size_t i = 0; i < tot; i++) fe_to_words(dig[i], tree + 4 * i);
    free(dig);
    return 0;
}

/* digest of one row of `len` Goldilocks values (contiguous) */
int orc_merkle16_leaf(const u64 *row, size_t len, u64 *out) {
    const p254_table *tb = &TAB[1];
    if (tb->t != 17) return -1;
    fe_to_words(leaf_of(tb, row, len, 1), out);
    return 0;
}
