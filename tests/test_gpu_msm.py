"""GPU tests of the BN254 MSM (N6): bit-exact against the definition-level oracle on small inputs,
group-law properties at larger sizes."""
import random

import numpy as np
import pytest

from oracle import naive_bn254 as B

pytestmark = pytest.mark.gpu


def limbs(v, n=8):
    return [(v >> (32 * k)) & 0xFFFFFFFF for k in range(n)]


@pytest.fixture(scope="module")
def table():
    rnd = random.Random(7)
    pts = [B.mul(B.G, rnd.randrange(1, B.R)) for _ in range(64)]
    assert all(B.on_curve(p) for p in pts)
    return pts


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 64, 200, 1000])
def test_msm_matches_oracle(prover, table, n):
    rnd = random.Random(100 + n)
    pts = [table[rnd.randrange(len(table))] for _ in range(n)]
    scs = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    for i, v in enumerate([0, 1, B.R - 1, B.R, (1 << 254) - 1, 1 << 253][:n]):
        scs[i] = v
    if n <= 200:
        assert prover.msm_bn254(pts, scs) == B.msm(pts, scs)
    else:       # (the checker's double-and-add over 1000 points is half a minute of Python: per DISTINCT point of the 64-point table, as below)
        by_pt = {}
        for p, sc in zip(pts, scs):
            by_pt[p] = (by_pt.get(p, 0) + sc) % B.R
        assert prover.msm_bn254(pts, scs) == B.msm(list(by_pt), list(by_pt.values()))


def test_msm_in_several_runs_matches_oracle(prover, table):
    """n above the run size: partial sums of the runs are added on the host (forced here with 2^6-point runs)"""
    rnd = random.Random(4711)
    n = 300
    pts = [table[rnd.randrange(len(table))] for _ in range(n)]
    scs = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    prover.set_tuning("msm_chunk_log", 6)
    try:
        assert prover.msm_bn254(pts, scs) == B.msm(pts, scs)
    finally:
        prover.set_tuning("msm_chunk_log", 0)


@pytest.mark.parametrize("kind", ["ones", "tiny", "two_values", "top_heavy"])
def test_msm_skewed_scalars_take_the_heavy_bucket_path(prover, table, kind):
    """non-uniform scalars put thousands of points into one bucket: such buckets are summed by whole workgroups"""
    rnd = random.Random(31 + len(kind))
    n = 3000
    pts = [table[rnd.randrange(len(table))] for _ in range(n)]
    if kind == "ones":
        scs = [1] * n
    elif kind == "tiny":
        scs = [rnd.randrange(0, 4) for _ in range(n)]
    elif kind == "two_values":
        scs = [rnd.choice([B.R - 1, 12345678901234567890]) for _ in range(n)]
    else:   # random low bits, identical top bits: every window above the first is one heavy bucket
        scs = [(0x2F << 248) | (0xABCDEF << 100) | rnd.randrange(0, 1 << 20) for _ in range(n)]
    # the checker's double-and-add over 3000 points took a minute per case in pure Python: the points come from a small table, so the same sum
    # is taken per DISTINCT point first (sum_i s_i P_(t_i) = sum_t (sum_{i: t_i = t} s_i mod r) P_t) -- the definition plus linearity
    by_pt = {}
    for p, sc in zip(pts, scs):
        by_pt[p] = (by_pt.get(p, 0) + sc) % B.R
    assert prover.msm_bn254(pts, scs) == B.msm(list(by_pt), list(by_pt.values()))


def test_msm_g2_heavy_bucket_path(prover, table_g2):
    rnd = random.Random(5)
    n = 700
    pts = [table_g2[rnd.randrange(len(table_g2))] for _ in range(n)]
    scs = [rnd.randrange(1, 3) for _ in range(n)]
    assert prover.msm_bn254_g2(pts, scs) == B.msm_g2(pts, scs)


def test_msm_infinity_inputs_and_cancellation(prover, table):
    p = table[0]
    neg = (p[0], B.Q - p[1])
    assert prover.msm_bn254([p, neg], [5, 5]) is None                      # P - P = infinity
    assert prover.msm_bn254([(0, 0), p], [9, 3]) == B.mul(p, 3)            # (0,0) encodes infinity
    assert prover.msm_bn254([p, p, p], [1, 1, 1]) == B.mul(p, 3)           # doubling path inside a bucket
    assert prover.msm_bn254([p], [0]) is None


def test_msm_same_point_many_times(prover):
    # every point equal: exercises the doubling branch of the bucket accumulation; answer = (sum s) * G
    n = 1 << 14
    rnd = np.random.default_rng(3)
    scs = rnd.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    scs[:, 7] &= 0x0FFFFFFF
    pts = np.zeros((n, 16), dtype=np.uint32)
    pts[:, 0], pts[:, 8] = 1, 2
    total = sum(sum(int(scs[i, k]) << (32 * k) for k in range(8)) for i in range(n)) % B.R
    assert prover.msm_bn254_arrays(pts, scs) == B.mul(B.G, total)


def test_msm_large_table_property(prover, table):
    # 2^17 terms over a 64-point table: expected = sum_j (sum_{i: idx_i = j} s_i) * P_j
    n = 1 << 17
    rnd = np.random.default_rng(5)
    idx = rnd.integers(0, len(table), size=n)
    scs = rnd.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    scs[:, 7] &= 0x0FFFFFFF
    tab = np.array([limbs(p[0]) + limbs(p[1]) for p in table], dtype=np.uint32)
    pts = tab[idx]
    ints = [sum(int(scs[i, k]) << (32 * k) for k in range(8)) for i in range(n)]
    per = [0] * len(table)
    for i, j in enumerate(idx):
        per[j] = (per[j] + ints[i]) % B.R
    assert prover.msm_bn254_arrays(pts, scs) == B.msm(table, per)


@pytest.fixture(scope="module")
def table_g2():
    rnd = random.Random(77)
    pts = [B.mul_g2(B.G2, rnd.randrange(1, B.R)) for _ in range(16)]
    assert B.on_curve_g2(B.G2) and all(B.on_curve_g2(p) for p in pts)
    return pts


@pytest.mark.parametrize("n", [0, 1, 2, 5, 64, 300])
def test_msm_g2_matches_oracle(prover, table_g2, n):
    rnd = random.Random(900 + n)
    pts = [table_g2[rnd.randrange(len(table_g2))] for _ in range(n)]
    scs = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    for i, v in enumerate([0, 1, B.R - 1, B.R][:n]):
        scs[i] = v
    if n > 4:
        pts[4] = None                                     # infinity among the inputs
    assert prover.msm_bn254_g2(pts, scs) == B.msm_g2(pts, scs)


def test_msm_g2_cancellation_and_doubling(prover, table_g2):
    p = table_g2[0]
    neg = (p[0], ((-p[1][0]) % B.Q, (-p[1][1]) % B.Q))
    assert prover.msm_bn254_g2([p, neg], [5, 5]) is None
    assert prover.msm_bn254_g2([p] * 7, [1] * 7) == B.mul_g2(p, 7)      # same point in one bucket: the doubling branch


def test_golden_msm_vectors_through_cabi(prover, golden):
    """double-and-add vectors (oracle/naive_bn254.py -> tests/golden/vectors.json): infinity, zero and r-1 scalars, P + (-P)"""
    for case in golden["msm_g1"]:
        pts = [(int(x), int(y)) for x, y in case["points"]]
        got = prover.msm_bn254(pts, [int(v) for v in case["scalars"]])
        want = (int(case["sum"][0]), int(case["sum"][1]))
        assert (got or (0, 0)) == want
    for case in golden["msm_g2"]:
        pts = [((int(p[0][0]), int(p[0][1])), (int(p[1][0]), int(p[1][1]))) for p in case["points"]]
        got = prover.msm_bn254_g2(pts, [int(v) for v in case["scalars"]])
        assert got == ((int(case["sum"][0][0]), int(case["sum"][0][1])), (int(case["sum"][1][0]), int(case["sum"][1][1])))
