/*
 * oracle/bn254_gen.c -- bulk generator of DISTINCT BN254 G1 points with a known discrete log (TEST INFRASTRUCTURE).
 *
 * Only tests/ and bench.py's input preparation use it: a 2^26-point multi-scalar multiplication (BASELINE.json
 * configs[4]) needs 2^26 different valid curve points to be a memory-realistic input, and a checkable answer:
 *     P_i = (start + i) * G          =>   sum_i s_i P_i = (sum_i s_i (start + i) mod r) * G
 * The points come out in the C-ABI layout of zp_msm_bn254 (include/zeth_prover.h): 16 little-endian u32 per point,
 * x then y, standard (non-Montgomery) form.  Affine chord additions against a table j*G with one batched inversion
 * per block of 1024 points; nothing is shared with the HIP kernels (29-bit-limb Jacobian arithmetic).
 * No reference counterpart: /root/reference holds no curve arithmetic.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef uint64_t u64;
typedef uint32_t u32;
typedef unsigned __int128 u128;
typedef struct { u64 v[4]; } fq;

static const fq QM = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};
static u64 QINV;      /* -q^-1 mod 2^64 */
static fq R1, R2;     /* 2^256 mod q, 2^512 mod q */
static int ready = 0;

static int geq(const fq *a, const fq *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->v[i] != b->v[i]) return a->v[i] > b->v[i];
    }
    return 1;
}
static void sub_nored(fq *r, const fq *a, const fq *b) {
    u64 br = 0;
    for (int i = 0; i < 4; i++) {
        u128 d = (u128)a->v[i] - b->v[i] - br;
        r->v[i] = (u64)d;
        br = (u64)(d >> 64) & 1;
    }
}
static void fq_add(fq *r, const fq *a, const fq *b) {
    u64 c = 0;
    fq t;
    for (int i = 0; i < 4; i++) {
        u128 s = (u128)a->v[i] + b->v[i] + c;
        t.v[i] = (u64)s;
        c = (u64)(s >> 64);
    }
    if (c || geq(&t, &QM)) sub_nored(&t, &t, &QM);
    *r = t;
}
static void fq_sub(fq *r, const fq *a, const fq *b) {
    fq t;
    if (geq(a, b)) sub_nored(&t, a, b);
    else { fq u; sub_nored(&u, b, a); sub_nored(&t, &QM, &u); }
    *r = t;
}
static void fq_mul(fq *r, const fq *a, const fq *b) {   /* Montgomery product a*b/2^256 mod q (CIOS) */
    u64 t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u64 c = 0;
        for (int j = 0; j < 4; j++) {
            u128 s = (u128)a->v[j] * b->v[i] + t[j] + c;
            t[j] = (u64)s;
            c = (u64)(s >> 64);
        }
        u128 s = (u128)t[4] + c;
        t[4] = (u64)s;
        t[5] = (u64)(s >> 64);
        u64 m = t[0] * QINV;
        s = (u128)m * QM.v[0] + t[0];
        c = (u64)(s >> 64);
        for (int j = 1; j < 4; j++) {
            s = (u128)m * QM.v[j] + t[j] + c;
            t[j - 1] = (u64)s;
            c = (u64)(s >> 64);
        }
        s = (u128)t[4] + c;
        t[3] = (u64)s;
        t[4] = t[5] + (u64)(s >> 64);
    }
    fq o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || geq(&o, &QM)) sub_nored(&o, &o, &QM);
    *r = o;
}
static void fq_inv(fq *r, const fq *a) {   /* a^(q-2) */
    fq e = QM, acc = R1, base = *a;
    e.v[0] -= 2;
    for (int i = 0; i < 256; i++) {
        if ((e.v[i / 64] >> (i % 64)) & 1) fq_mul(&acc, &acc, &base);
        fq_mul(&base, &base, &base);
    }
    *r = acc;
}
static void setup(void) {
    if (ready) return;
    u64 x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - QM.v[0] * x;   /* Newton: x = q^-1 mod 2^64 */
    QINV = (u64)0 - x;
    fq one = {{1, 0, 0, 0}}, t = one;
    for (int i = 0; i < 512; i++) {
        fq_add(&t, &t, &t);
        if (i == 255) R1 = t;
    }
    R2 = t;
    ready = 1;
}

typedef struct { fq x, y; } pt;   /* affine, Montgomery form; never infinity in this file */

static void pt_add(pt *r, const pt *p, const pt *q) {   /* general affine addition incl. doubling (p != -q) */
    fq num, den, lam, x3, y3, t;
    if (memcmp(&p->x, &q->x, sizeof(fq)) == 0) {
        fq_mul(&t, &p->x, &p->x);
        fq_add(&num, &t, &t);
        fq_add(&num, &num, &t);
        fq_add(&den, &p->y, &p->y);
    } else {
        fq_sub(&num, &q->y, &p->y);
        fq_sub(&den, &q->x, &p->x);
    }
    fq_inv(&den, &den);
    fq_mul(&lam, &num, &den);
    fq_mul(&x3, &lam, &lam);
    fq_sub(&x3, &x3, &p->x);
    fq_sub(&x3, &x3, &q->x);
    fq_sub(&t, &p->x, &x3);
    fq_mul(&y3, &lam, &t);
    fq_sub(&y3, &y3, &p->y);
    r->x = x3;
    r->y = y3;
}
static void pt_mul_g(pt *r, const pt *g, u64 k) {   /* k >= 1 */
    pt acc = *g, base = *g;
    int started = 0;
    for (int i = 0; i < 64; i++) {
        if ((k >> i) & 1) {
            if (!started) { acc = base; started = 1; }
            else pt_add(&acc, &acc, &base);
        }
        pt_add(&base, &base, &base);
    }
    *r = acc;
}
static void store(u32 *out, const pt *p) {
    fq one = {{1, 0, 0, 0}}, x, y;
    fq_mul(&x, &p->x, &one);
    fq_mul(&y, &p->y, &one);
    for (int i = 0; i < 4; i++) {
        out[2 * i] = (u32)x.v[i]; out[2 * i + 1] = (u32)(x.v[i] >> 32);
        out[8 + 2 * i] = (u32)y.v[i]; out[8 + 2 * i + 1] = (u32)(y.v[i] >> 32);
    }
}

#define BLK 1024
/* out: n points of 16 u32; point i = (start + i) * G.  start must be > BLK (keeps every chord addition a chord). */
int orc_bn254_consecutive_points(u32 *out, size_t n, u64 start) {
    setup();
    if (start <= BLK || start + n < start) return -1;
    pt g;
    fq one = {{1, 0, 0, 0}}, two = {{2, 0, 0, 0}};
    fq_mul(&g.x, &one, &R2);
    fq_mul(&g.y, &two, &R2);
    pt *tab = (pt *)malloc((BLK + 1) * sizeof(pt));   /* tab[j] = j*G, j = 1..BLK */
    tab[1] = g;
    for (int j = 2; j <= BLK; j++) pt_add(&tab[j], &tab[j - 1], &g);
    const size_t nblk = (n + BLK - 1) / BLK;
#pragma omp parallel
    {
        fq *den = (fq *)malloc(BLK * sizeof(fq)), *pre = (fq *)malloc(BLK * sizeof(fq));
        pt base;
        size_t have = (size_t)-1;
#pragma omp for schedule(static)
        for (size_t b = 0; b < nblk; b++) {
            if (have != b) pt_mul_g(&base, &g, start + b * BLK);   /* first block of this thread's range */
            const size_t cnt = (b + 1) * BLK <= n ? BLK : n - b * BLK;
            u32 *o = out + b * BLK * 16;
            store(o, &base);
            /* den[j] = x(T_j) - x(base), j = 1..BLK (the last one yields the next base); one inversion for all */
            fq acc = R1;
            for (int j = 1; j <= BLK; j++) {
                fq_sub(&den[j - 1], &tab[j].x, &base.x);
                pre[j - 1] = acc;
                fq_mul(&acc, &acc, &den[j - 1]);
            }
            fq inv;
            fq_inv(&inv, &acc);
            pt nextbase = base;
            for (int j = BLK; j >= 1; j--) {
                fq dinv, num, lam, x3, y3, t;
                fq_mul(&dinv, &inv, &pre[j - 1]);
                fq_mul(&inv, &inv, &den[j - 1]);
                if ((size_t)j >= cnt && j != BLK) continue;
                fq_sub(&num, &tab[j].y, &base.y);
                fq_mul(&lam, &num, &dinv);
                fq_mul(&x3, &lam, &lam);
                fq_sub(&x3, &x3, &base.x);
                fq_sub(&x3, &x3, &tab[j].x);
                fq_sub(&t, &base.x, &x3);
                fq_mul(&y3, &lam, &t);
                fq_sub(&y3, &y3, &base.y);
                pt r = {x3, y3};
                if (j == BLK) nextbase = r;
                if ((size_t)j < cnt) store(o + (size_t)j * 16, &r);
            }
            base = nextbase;
            have = b + 1;
        }
        free(den);
        free(pre);
    }
    free(tab);
    return 0;
}

/* out[0..9] (little-endian u32 limbs of a 320-bit integer) = sum_i scalar_i * (start + i), scalars 8 x u32 each */
void orc_bn254_weighted_scalar_sum(const u32 *scalars, size_t n, u64 start, u32 *out10) {
    u64 tot[6] = {0, 0, 0, 0, 0, 0};
#pragma omp parallel
    {
        u64 loc[6] = {0, 0, 0, 0, 0, 0};
#pragma omp for schedule(static)
        for (size_t i = 0; i < n; i++) {
            const u64 w = start + i;
            u64 c = 0;
            for (int k = 0; k < 4; k++) {
                const u64 limb = (u64)scalars[i * 8 + 2 * k] | ((u64)scalars[i * 8 + 2 * k + 1] << 32);
                u128 s = (u128)limb * w + loc[k] + c;
                loc[k] = (u64)s;
                c = (u64)(s >> 64);
            }
            u128 s = (u128)loc[4] + c;
            loc[4] = (u64)s;
            loc[5] += (u64)(s >> 64);
        }
#pragma omp critical
        {
            u64 c = 0;
            for (int k = 0; k < 6; k++) {
                u128 s = (u128)tot[k] + loc[k] + c;
                tot[k] = (u64)s;
                c = (u64)(s >> 64);
            }
        }
    }
    for (int k = 0; k < 5; k++) { out10[2 * k] = (u32)tot[k]; out10[2 * k + 1] = (u32)(tot[k] >> 32); }
}
