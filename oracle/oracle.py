"""oracle/oracle.py -- ctypes binding of the CPU restatement (TEST INFRASTRUCTURE ONLY).

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (eigen_zeth_amd) must never import this module.  PARITY UNPINNED -- see gl_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")

P = 0xFFFFFFFF00000001
ROOT32_DEFAULT = 1753635133440165772
ROOT32_ALT = 7277203076849721926
SHIFT_DEFAULT = 49

_u64p = C.POINTER(C.c_uint64)


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(
            os.path.getmtime(os.path.join(_HERE, f)) for f in ("gl_oracle.c", "bn254_gen.c", "bn254_hash.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def _cpu_budget():
    """threads the OpenMP loops of the checker should use: the CPUs this process may run on, capped by the CFS quota of its
    cgroup.  A container can SEE every CPU of a large host (256 on the GPU boxes) and be throttled to 16 of them; a 256-thread
    team with spinning barriers on a 16-CPU quota ran some checks 20-60 x slower, and erratically, from one box to the next."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    paths = ["/sys/fs/cgroup/cpu.max"]
    try:
        for line in open("/proc/self/cgroup"):
            paths.insert(0, "/sys/fs/cgroup" + line.strip().split(":", 2)[-1].rstrip("/") + "/cpu.max")
    except OSError:
        pass
    quota = None
    for pth in paths:
        try:
            q, per = open(pth).read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
                break
        except (OSError, ValueError):
            continue
    if quota is None:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            if q > 0:
                quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")      # idle workers sleep instead of spinning beside the GPU's host threads
        if "OMP_NUM_THREADS" not in os.environ:
            os.environ["OMP_NUM_THREADS"] = str(_cpu_budget())
        L = C.CDLL(_SO)
        u64, i32, sz = C.c_uint64, C.c_int, C.c_size_t
        for name in ("orc_add", "orc_sub", "orc_mul", "orc_pow"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64, u64]
        L.orc_inv.restype = u64
        L.orc_inv.argtypes = [u64]
        L.orc_root.restype = u64
        L.orc_root.argtypes = [u64, i32]
        L.orc_ntt.argtypes = [_u64p, i32, i32, u64]
        L.orc_intt.argtypes = [_u64p, i32, i32, u64]
        L.orc_lde.argtypes = [_u64p, _u64p, i32, i32, i32, u64, u64]
        L.orc_poseidon_perm.argtypes = [_u64p, sz, _u64p, _u64p]
        L.orc_poseidon_trace.argtypes = [_u64p, sz, _u64p, _u64p, _u64p, _u64p]
        L.orc_merkle_commit.argtypes = [_u64p, sz, i32, _u64p, _u64p, _u64p]
        L.orc_merkle_commit_rows.argtypes = [_u64p, sz, sz, _u64p, _u64p, _u64p]
        L.orc_linear_hash.argtypes = [_u64p, sz, _u64p, _u64p, _u64p]
        L.orc_merkle_path.argtypes = [_u64p, sz, sz, _u64p]
        L.orc_merkle_verify.restype = i32
        L.orc_merkle_verify.argtypes = [_u64p, sz, sz, _u64p, _u64p, _u64p, _u64p]
        L.orc_e3_mul.argtypes = [_u64p, _u64p, _u64p]
        L.orc_e3_pow.argtypes = [_u64p, _u64p, _u64p]
        L.orc_e3_inv.argtypes = [_u64p, _u64p]
        L.orc_deep_quotient.argtypes = [_u64p, i32, _u64p, i32, i32, i32, _u64p, _u64p, _u64p, _u64p, _u64p, u64, u64, _u64p]
        L.orc_poly_eval_e3_cols.argtypes = [_u64p, sz, i32, _u64p, _u64p]
        L.orc_deep_quotient_fast.argtypes = L.orc_deep_quotient.argtypes
        L.orc_grand_product.argtypes = [_u64p, _u64p, sz, _u64p, _u64p]
        L.orc_logup_columns.argtypes = [_u64p, _u64p, _u64p, sz, _u64p, _u64p]
        L.orc_quotient_program.restype = i32
        L.orc_quotient_program.argtypes = [_u64p, sz, _u64p, _u64p, sz, sz, _u64p, _u64p, _u64p, u64, u64, u64, _u64p]
        L.orc_bn254_consecutive_points.restype = i32
        L.orc_bn254_consecutive_points.argtypes = [C.c_void_p, sz, u64]
        L.orc_bn254_weighted_scalar_sum.argtypes = [C.c_void_p, sz, u64, C.c_void_p]
        L.orc_coset_scale.argtypes = [_u64p, sz, i32, u64]
        L.orc_quotient_program_rows.restype = i32
        L.orc_quotient_program_rows.argtypes = [_u64p, sz, _u64p, sz, _u64p, sz, sz, sz, sz, sz, _u64p, _u64p, _u64p, u64, u64, u64, _u64p, sz]
        L.orc_deep_quotient_rows.argtypes = [_u64p, i32, sz, _u64p, i32, sz, i32, sz, sz, i32, _u64p, _u64p, _u64p, _u64p, _u64p, u64, u64, _u64p, sz]
        L.orc_pow_grind.restype = u64
        L.orc_pow_grind.argtypes = [_u64p, i32, _u64p, _u64p]
        L.orc_fri_fold.argtypes = [_u64p, _u64p, i32, i32, _u64p, u64, u64]
        L.orc_poly_eval.restype = u64
        L.orc_poly_eval.argtypes = [_u64p, sz, u64]
        L.orc_poly_eval_e3.argtypes = [_u64p, sz, _u64p, _u64p]
        L.orc_set_simple_ntt.argtypes = [i32]
        L.orc_num_threads.restype = i32
        L.orc_set_threads.argtypes = [i32]
        L.orc_p254_set.restype = i32
        L.orc_p254_set.argtypes = [i32, i32, _u64p, _u64p]
        L.orc_p254_perm.restype = i32
        L.orc_p254_perm.argtypes = [_u64p, sz, i32]
        L.orc_merkle16_nodes.restype = sz
        L.orc_merkle16_nodes.argtypes = [sz]
        L.orc_merkle16_tree.restype = i32
        L.orc_merkle16_tree.argtypes = [_u64p, sz, i32, _u64p]
        L.orc_merkle16_leaf.restype = i32
        L.orc_merkle16_leaf.argtypes = [_u64p, sz, _u64p]
        _lib = L
    return _lib


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def ntt(cols, root32=ROOT32_DEFAULT):
    """cols: uint64 [W][N] -> new array, natural in / natural out"""
    a = _arr(cols).copy()
    W, N = a.shape
    lib().orc_ntt(_p(a), N.bit_length() - 1, W, root32)
    return a


def intt(cols, root32=ROOT32_DEFAULT):
    a = _arr(cols).copy()
    W, N = a.shape
    lib().orc_intt(_p(a), N.bit_length() - 1, W, root32)
    return a


def lde(cols, logb, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    a = _arr(cols)
    W, N = a.shape
    out = np.empty((W, N << logb), dtype=np.uint64)
    lib().orc_lde(_p(a), _p(out), N.bit_length() - 1, logb, W, shift, root32)
    return out


def poseidon_perm(states, rc, mds):
    """states: uint64 [B][12] -> permuted copy"""
    s = _arr(states).copy()
    rc = _arr(rc)
    mds = _arr(mds)
    lib().orc_poseidon_perm(_p(s), s.shape[0], _p(rc), _p(mds))
    return s


def poseidon_trace(inputs, rc, mds):
    """inputs: uint64 [B][12] -> (states [12][32 B], cubes [12][32 B]): the round-by-round witness of a Poseidon AIR"""
    a = _arr(inputs)
    B = a.shape[0]
    st = np.empty((12, 32 * B), dtype=np.uint64)
    cu = np.empty((12, 32 * B), dtype=np.uint64)
    lib().orc_poseidon_trace(_p(a), B, _p(st), _p(cu), _p(_arr(rc)), _p(_arr(mds)))
    return st, cu


def merkle_commit(cols, rc, mds):
    """cols: uint64 [W][M] -> tree uint64 [(2M-1)][4] (leaves first, root last)"""
    a = _arr(cols)
    W, M = a.shape
    tree = np.empty((2 * M - 1, 4), dtype=np.uint64)
    lib().orc_merkle_commit(_p(a), M, W, _p(tree), _p(_arr(rc)), _p(_arr(mds)))
    return tree


def merkle_commit_rows(rows, rc, mds):
    a = _arr(rows)
    M, ln = a.shape
    tree = np.empty((2 * M - 1, 4), dtype=np.uint64)
    lib().orc_merkle_commit_rows(_p(a), M, ln, _p(tree), _p(_arr(rc)), _p(_arr(mds)))
    return tree


def linear_hash(row, rc, mds):
    r = _arr(row)
    out = np.empty(4, dtype=np.uint64)
    lib().orc_linear_hash(_p(r), r.shape[0], _p(out), _p(_arr(rc)), _p(_arr(mds)))
    return out


def merkle_path(tree, idx):
    t = _arr(tree)
    M = (t.shape[0] + 1) // 2
    depth = M.bit_length() - 1
    path = np.empty((max(depth, 1), 4), dtype=np.uint64)
    lib().orc_merkle_path(_p(t), M, idx, _p(path))
    return path[:depth]


def merkle_verify(leaf4, M, idx, path, root, rc, mds):
    pth = _arr(path) if len(path) else np.zeros((1, 4), dtype=np.uint64)
    return bool(lib().orc_merkle_verify(_p(_arr(leaf4)), M, idx, _p(pth), _p(_arr(root)),
                                        _p(_arr(rc)), _p(_arr(mds))))


def e3_mul(a, b):
    out = np.empty(3, dtype=np.uint64)
    lib().orc_e3_mul(_p(_arr(a)), _p(_arr(b)), _p(out))
    return out


def e3_inv(a):
    e = P ** 3 - 2
    limbs = np.array([(e >> (64 * i)) & (2 ** 64 - 1) for i in range(3)], dtype=np.uint64)
    out = np.empty(3, dtype=np.uint64)
    lib().orc_e3_pow(_p(_arr(a)), _p(limbs), _p(out))
    return out


def deep_quotient(cols_a, cols_b, n_next, z, zw, gamma, ev_z, ev_zw, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT, fast=False):
    a = _arr(cols_a)
    Wa, M = a.shape
    b = _arr(cols_b) if cols_b is not None and len(cols_b) else np.zeros((1, M), dtype=np.uint64)
    Wb = 0 if cols_b is None or len(cols_b) == 0 else b.shape[0]
    out = np.empty((3, M), dtype=np.uint64)
    ezw = _arr(ev_zw) if n_next else np.zeros((1, 3), dtype=np.uint64)
    (lib().orc_deep_quotient_fast if fast else lib().orc_deep_quotient)(_p(a), Wa, _p(b), Wb, M.bit_length() - 1, n_next, _p(_arr(z)), _p(_arr(zw)),
                            _p(_arr(gamma)), _p(_arr(ev_z)), _p(ezw), shift, root32, _p(out))
    return out


def grand_product(a, b, gamma):
    a, b = _arr(a), _arr(b)
    out = np.empty((3, a.shape[0]), dtype=np.uint64)
    lib().orc_grand_product(_p(a), _p(b), a.shape[0], _p(_arr(gamma)), _p(out))
    return out


def logup_columns(a, t, m, gamma):
    a, t, m = _arr(a), _arr(t), _arr(m)
    out = np.empty((9, a.shape[0]), dtype=np.uint64)
    lib().orc_logup_columns(_p(a), _p(t), _p(m), a.shape[0], _p(_arr(gamma)), _p(out))
    return out


def fri_fold(planes, logf, beta, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    """planes: uint64 [3][n] -> uint64 [3][n >> logf]"""
    a = _arr(planes)
    n = a.shape[1]
    out = np.empty((3, n >> logf), dtype=np.uint64)
    lib().orc_fri_fold(_p(a), _p(out), n.bit_length() - 1, logf, _p(_arr(beta)), shift, root32)
    return out


def poly_eval(coef, x):
    c = _arr(coef)
    return int(lib().orc_poly_eval(_p(c), c.shape[0], x))


def poly_eval_e3(coef, x3):
    c = _arr(coef)
    out = np.empty(3, dtype=np.uint64)
    lib().orc_poly_eval_e3(_p(c), c.shape[0], _p(_arr(x3)), _p(out))
    return out


def poly_eval_e3_cols(coef, x3):
    c = _arr(coef)
    W, n = c.shape
    out = np.empty((W, 3), dtype=np.uint64)
    lib().orc_poly_eval_e3_cols(_p(c), n, W, _p(_arr(x3)), _p(out))
    return out


def coset_scaled_coefficients(cols, shift=SHIFT_DEFAULT, root32=ROOT32_DEFAULT):
    """coefficients of the interpolant composed with the coset shift: iNTT(cols)[i] * shift^i"""
    a = intt(cols, root32)
    lib().orc_coset_scale(_p(a), a.shape[1], a.shape[0], shift)
    return a


def pow_grind(seed4, bits, rc, mds):
    """smallest nonce with Poseidon(seed || nonce || 0^7)[0] >> (64 - bits) == 0"""
    sd = _arr([int(v) for v in seed4])
    return int(lib().orc_pow_grind(_p(sd), int(bits), _p(_arr(rc)), _p(_arr(mds))))


def bn254_consecutive_points(n, start=1025):
    """uint32 [n][16]: point i = (start + i) * G of BN254 G1 in the zp_msm_bn254 layout (x then y, little-endian limbs)"""
    out = np.empty((n, 16), dtype=np.uint32)
    if lib().orc_bn254_consecutive_points(out.ctypes.data, n, start) != 0:
        raise ValueError("start must exceed 1024")
    return out


def bn254_weighted_scalar_sum(scalars, start=1025):
    """sum_i scalar_i * (start + i) as a Python int; scalars uint32 [n][8] little-endian limbs"""
    s = np.ascontiguousarray(scalars, dtype=np.uint32)
    out = np.zeros(10, dtype=np.uint32)
    lib().orc_bn254_weighted_scalar_sum(s.ctypes.data, s.shape[0], start, out.ctypes.data)
    return sum(int(out[k]) << (32 * k) for k in range(10))


def set_simple_ntt(on):
    """force the plain radix-2 loop for every size (the default switches to the cache-blocked form at 2^16)"""
    lib().orc_set_simple_ntt(int(bool(on)))


def num_threads():
    return int(lib().orc_num_threads())


def set_threads(n):
    lib().orc_set_threads(int(n))


def random_field(shape, seed):
    """uniform in [0,p) by rejection, PCG64(seed)  (SURVEY.md 8d synthetic inputs)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 2 ** 64, size=shape, dtype=np.uint64, endpoint=False)
    bad = a >= np.uint64(P)
    while bad.any():
        a[bad] = rng.integers(0, 2 ** 64, size=int(bad.sum()), dtype=np.uint64, endpoint=False)
        bad = a >= np.uint64(P)
    return a


# ---- BN128-hash mode (oracle/bn254_hash.c): field elements are Python ints < r here, 4-word arrays at the C boundary
def _fr_words(vals):
    a = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for k in range(4):
            a[i, k] = (v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return a


def _fr_ints(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, k]) << (64 * k) for k in range(4)) for i in range(a.shape[0])]


def p254_set(t, rp, rc, mds):
    """install the tables of width t (3 or 17): rc = (8 + rp) * t ints round-major, mds = t x t (nested lists or flat)"""
    flat = [v for row in mds for v in row] if isinstance(mds[0], (list, tuple)) else list(mds)
    rcw, mw = _fr_words(rc), _fr_words(flat)
    if lib().orc_p254_set(t, rp, _p(rcw), _p(mw)) != 0:
        raise ValueError("bad Poseidon-BN254 tables")


def p254_perm(states, t):
    """list of states (t ints each) -> permuted states"""
    a = _fr_words([v for st in states for v in st])
    if lib().orc_p254_perm(_p(a), len(states), t) != 0:
        raise ValueError("Poseidon-BN254 tables of width %d not installed" % t)
    out = _fr_ints(a)
    return [out[i * t:(i + 1) * t] for i in range(len(states))]


def merkle16_nodes(M):
    return int(lib().orc_merkle16_nodes(M))


def merkle16_tree(cols):
    """cols uint64 [W][M] -> uint64 [nodes][4]: leaves, then each level, root last"""
    a = _arr(cols)
    W, M = a.shape
    tree = np.empty((merkle16_nodes(M), 4), dtype=np.uint64)
    if lib().orc_merkle16_tree(_p(a), M, W, _p(tree)) != 0:
        raise ValueError("Poseidon-BN254 tables (t = 17) not installed")
    return tree


def merkle16_leaf(row):
    a = _arr(row).reshape(-1)
    out = np.empty(4, dtype=np.uint64)
    if lib().orc_merkle16_leaf(_p(a), a.size, _p(out)) != 0:
        raise ValueError("Poseidon-BN254 tables (t = 17) not installed")
    return _fr_ints(out)[0]
