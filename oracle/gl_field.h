/* oracle/gl_field.h -- field helpers for generated CPU constraint code (TEST INFRASTRUCTURE;
 * same role and same caveats as gl_oracle.c: CPU restatement, parity with the reference unpinned). */
#ifndef GL_FIELD_H
#define GL_FIELD_H
#include <stdint.h>
typedef uint64_t u64;
typedef unsigned __int128 u128;
#define GL_P 0xFFFFFFFF00000001ULL
#define GL_EPS 0xFFFFFFFFULL
static inline u64 gl_add(u64 a, u64 b) { u64 s = a + b; if (s < a) s += GL_EPS; if (s >= GL_P) s -= GL_P; return s; }
static inline u64 gl_sub(u64 a, u64 b) { u64 d = a - b; if (a < b) d -= GL_EPS; return d; }
static inline u64 gl_mul(u64 a, u64 b) {
    u128 x = (u128)a * b;
    u64 lo = (u64)x, hi = (u64)(x >> 64), hh = hi >> 32, hl = hi & GL_EPS;
    u64 t0 = lo - hh; if (lo < hh) t0 -= GL_EPS;
    u64 t1 = hl * GL_EPS, r = t0 + t1; if (r < t1) r += GL_EPS; if (r >= GL_P) r -= GL_P;
    return r;
}
#endif
