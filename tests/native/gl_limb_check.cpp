// Host check of eigen_zeth_amd/csrc/gl_limb.hpp (test infrastructure): the limb-form add / sub / rotations / x 2^(12 e) / product / canonicalisation
// against big-integer arithmetic mod p on random and edge values.  Built and run by tests/test_gl_limb.py (g++, no GPU).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "gl_limb.hpp"
typedef __int128 i128;
static u64 val(const gl_l4 &x) {   // limbs -> canonical by big-integer arithmetic
    i128 v = 0;
    for (int i = 3; i >= 0; i--) v = v * (i128)(1 << 24) + (i128)(i32)x.l[i];
    i128 m = v % (i128)GL_P;
    if (m < 0) m += GL_P;
    return (u64)m;
}
static u64 mulmod(u64 a, u64 b) { return (u64)(((unsigned __int128)a * b) % GL_P); }
int main() {
    std::mt19937_64 g(1);
    u64 edge[] = {0, 1, 2, GL_P - 1, GL_P - 2, 0xFFFFFFFFULL, 0x100000000ULL, 0xFFFFFFFF00000000ULL, 0x7FFFFFFF00000000ULL, 0x7FFFFFFEFFFFFFFFULL, 0x8000000000000000ULL, 0x7FFFFFFFFFFFFFFFULL, 0xFFFFFFFEFFFFFFFFULL, 0x0000000100000001ULL, 0xFFFFFFULL, 0x1000000ULL};
    const int NE = sizeof(edge) / sizeof(edge[0]);
    size_t bad = 0, n = 0;
    auto pick = [&](int i) { return i < NE ? edge[i] : g() % GL_P; };
    for (int i = 0; i < 3000; i++)
        for (int j = 0; j < 300; j++) {
            const u64 x = pick(i), w = pick(j), y = pick((i * 7 + j) % 3000);
            gl_l4 a = gl_l4_from(x), b = gl_l4_from(y);
            bad += val(a) != x;
            // a few levels of growth: s = 15 x - y style combos stay below 2^28
            gl_l4 s = gl_l4_add(a, b), d = gl_l4_sub(a, b);
            for (int k = 0; k < 3; k++) { gl_l4 s2 = gl_l4_add(s, d), d2 = gl_l4_sub(s, d); s = s2; d = gl_l4_rot<1>(d2); }
            const u64 sv = val(s), dv = val(d);
            const gl_w4 W = gl_l4_factor(w);
            bad += gl_l4_mul(s, W) != mulmod(sv, w);
            bad += gl_l4_mul(d, W) != mulmod(dv, w);
            bad += gl_l4_canon(d) != dv;
            bad += gl_l4_canon(s) != sv;
            bad += gl_l4_canon(a) != x;
            bad += val(gl_l4_mul_c16<1>(d)) != mulmod(dv, 1ULL << 12);
            bad += val(gl_l4_mul_c16<2>(d)) != mulmod(dv, 1ULL << 24);
            bad += val(gl_l4_mul_c16<3>(d)) != mulmod(dv, 1ULL << 36);
            bad += val(gl_l4_mul_c16<4>(d)) != mulmod(dv, 1ULL << 48);
            bad += val(gl_l4_mul_c16<5>(d)) != mulmod(dv, 1ULL << 60);
            bad += val(gl_l4_mul_c16<6>(d)) != mulmod(mulmod(dv, 1ULL << 36), 1ULL << 36);
            bad += val(gl_l4_mul_c16<7>(d)) != mulmod(mulmod(dv, 1ULL << 42), 1ULL << 42);
            for (int k = 0; k < 4; k++) { if ((i32)d.l[k] >= (1 << 28) || (i32)d.l[k] <= -(1 << 28)) bad++; }
            n++;
        }
    printf("%zu cases, %zu mismatches\n", n, bad);
    return bad != 0;
}
