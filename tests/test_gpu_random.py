"""Randomised GPU parity sweep (-m gpu): seeded random shapes and parameters for every C-ABI compute entry point,
each compared bit-for-bit with the CPU restatement.  Complements the fixed cases of test_gpu_parity.py /
test_gpu_stark.py / test_gpu_msm.py: odd widths, every blow-up, both roots of unity, custom coset shifts, every MSM
window width."""
import random

import numpy as np
import pytest

from oracle import naive_bn254 as B
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _cases(seed, n):
    rnd = random.Random(seed)
    return [rnd.randrange(1 << 30) for _ in range(n)]


@pytest.mark.parametrize("seed", _cases(1, 10))
def test_random_ntt_lde(prover, seed):
    from eigen_zeth_amd import native
    rnd = random.Random(seed)
    logn, W, logb = rnd.randrange(0, 15), rnd.randrange(1, 10), rnd.randrange(0, 5)
    root = rnd.choice([O.ROOT32_DEFAULT, O.ROOT32_ALT])
    shift = rnd.choice([49, 7, 3 ** 20 % O.P, O.P - 2])
    x = O.random_field((W, 1 << logn), seed)
    try:
        prover.set_constants(native.ZP_CONST_ROOT32, [root])
        d = prover.upload(x)
        o = prover.alloc(W << logn)
        prover.ntt(d, o, logn, W)
        assert (prover.download(o, x.shape) == O.ntt(x, root)).all(), (logn, W, root)
        prover.intt(o, o, logn, W)
        assert (prover.download(o, x.shape) == x).all()
        e = prover.alloc(W << (logn + logb))
        prover.lde(d, e, logn, logb, W, shift)
        assert (prover.download(e, (W, 1 << (logn + logb))) == O.lde(x, logb, shift, root)).all(), (logn, logb, W, shift, root)
    finally:
        prover.set_constants(native.ZP_CONST_ROOT32, [native.ROOT32_DEFAULT])


@pytest.mark.parametrize("seed", _cases(2, 10))
def test_random_merkle_and_openings(prover, tables, seed):
    rc, mds = tables
    rnd = random.Random(seed)
    M, W = 1 << rnd.randrange(0, 14), rnd.randrange(1, 41)
    cols = O.random_field((W, M), seed)
    ref = O.merkle_commit(cols, rc, mds)
    t = prover.alloc((2 * M - 1) * 4)
    prover.merkle_commit(prover.upload(cols), M, W, t)
    assert (prover.download(t, ref.shape) == ref).all(), (M, W)
    rows = np.ascontiguousarray(cols.T)
    prover.merkle_commit_rows(prover.upload(rows), M, W, t)
    assert (prover.download(t, ref.shape) == ref).all(), (M, W, "rows")
    idx = [rnd.randrange(M) for _ in range(5)]
    paths = prover.merkle_open_batch(t, M, idx)
    for i, j in enumerate(idx):
        assert (paths[i] == O.merkle_path(ref, j)).all()


@pytest.mark.parametrize("seed", _cases(3, 8))
def test_random_fri_fold_and_extension_kernels(prover, seed):
    rnd = random.Random(seed)
    logf = rnd.randrange(1, 5)
    logn = rnd.randrange(logf, 15)
    shift = rnd.choice([49, 49 ** 8 % O.P, 5])
    planes = O.random_field((3, 1 << logn), seed)
    beta = O.random_field((3,), seed + 1).tolist()
    o = prover.alloc(3 << (logn - logf))
    prover.fri_fold(prover.upload(planes), o, logn, logf, beta, shift)
    assert (prover.download(o, (3, 1 << (logn - logf))) == O.fri_fold(planes, logf, beta, shift)).all(), (logn, logf, shift)
    # OOD evaluation of random coefficient columns at a random extension point
    W = rnd.randrange(1, 7)
    coef = O.random_field((W, 1 << logn), seed + 2)
    z = O.random_field((3,), seed + 3).tolist()
    got = prover.poly_eval_ext(prover.upload(coef), logn, W, z)
    assert (np.asarray(got, dtype=np.uint64) == O.poly_eval_e3_cols(coef, z)).all(), (logn, W)
    # stage-2 witness columns
    n = rnd.randrange(1, 5000)
    a = O.random_field((n,), seed + 4)
    b = a[np.random.default_rng(seed).permutation(n)]
    g = O.random_field((3,), seed + 5).tolist()
    dz = prover.alloc(3 * n)
    prover.grand_product(prover.upload(a), prover.upload(b), n, g, dz)
    assert (prover.download(dz, (3, n)) == O.grand_product(a, b, g)).all(), n
    m = O.random_field((n,), seed + 6) % np.uint64(7)
    dl = prover.alloc(9 * n)
    prover.logup_columns(prover.upload(a), prover.upload(b), prover.upload(m), n, g, dl)
    assert (prover.download(dl, (9, n)) == O.logup_columns(a, b, m, g)).all(), n


@pytest.fixture(scope="module")
def g1_table():
    rnd = random.Random(99)
    return [B.mul(B.G, rnd.randrange(1, B.R)) for _ in range(24)]


@pytest.mark.parametrize("c", [6, 7, 9, 10, 11, 13, 16])
def test_msm_every_window_width(prover, g1_table, c):
    """window widths 6..16 (a sample) force every coarse/fine split of the index sort (hi = 0..6 bits) and partial top windows"""
    rnd = random.Random(1000 + c)
    n = rnd.randrange(260, 300)   # > MSM_HEAVY so that skewed top windows also take the heavy path
    pts = [g1_table[rnd.randrange(len(g1_table))] for _ in range(n)]
    scs = [rnd.randrange(0, B.R) for _ in range(n)]
    prover.set_tuning("msm_c", c)
    try:
        by_pt = {}          # the definition plus linearity: one double-and-add per DISTINCT point of the 24-point table
        for p, sc in zip(pts, scs):
            by_pt[p] = (by_pt.get(p, 0) + sc) % B.R
        assert prover.msm_bn254(pts, scs) == B.msm(list(by_pt), list(by_pt.values())), c
    finally:
        prover.set_tuning("msm_c", 0)
