"""zp_fixed_base_mul_bn254(_g2) -- the group elements of a Groth16 key from its scalars (csrc/msm.hip: 8-bit window table of one base point,
32 table additions per scalar on the GPU, batched inversions on host threads) -- against the double-and-add definition
(oracle/naive_bn254.py) and, at size, against the library's own MSM: sum_i w_i (s_i G) = (sum_i w_i s_i) G."""
import numpy as np
import pytest

from eigen_zeth_amd import native
from oracle import naive_bn254 as B

pytestmark = pytest.mark.gpu
G1W = np.array([1] + [0] * 7 + [2] + [0] * 7, dtype=np.uint32)


def _words(v, n):
    return [(v >> (32 * k)) & 0xFFFFFFFF for k in range(n)]


def _g2_words():
    (x0, x1), (y0, y1) = B.G2
    return np.array(_words(x0, 8) + _words(x1, 8) + _words(y0, 8) + _words(y1, 8), dtype=np.uint32)


def _pt1(row):
    x = sum(int(row[k]) << (32 * k) for k in range(8))
    y = sum(int(row[8 + k]) << (32 * k) for k in range(8))
    return None if x == 0 and y == 0 else (x, y)


def _pt2(row):
    v = [sum(int(row[8 * c + k]) << (32 * k) for k in range(8)) for c in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def test_fixed_base_g1_and_g2_match_double_and_add(prover):
    rng = np.random.default_rng(5)
    scal = [0, 1, 2, 255, 256, 257, B.R - 1, (1 << 253) + 12345] + [int.from_bytes(rng.bytes(32), "little") % B.R for _ in range(24)]
    sw = native.fr_words(scal)
    p1 = prover.fixed_base_mul(G1W, sw)
    p2 = prover.fixed_base_mul(_g2_words(), sw, g2=True)
    for i, s in enumerate(scal):
        assert _pt1(p1[i]) == (B.mul((1, 2), s) if s else None), i
        assert _pt2(p2[i]) == (B.mul_g2(B.G2, s) if s else None), i


def test_fixed_base_at_size_against_the_msm(prover):
    n = 1 << 16
    rng = np.random.default_rng(6)
    s = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
    sw = np.zeros((n, 4), dtype=np.uint64)
    sw[:, 0] = s
    sw[:, 2] = s >> np.uint64(7)                      # 190-bit scalars
    sw[5] = 0
    pts = prover.fixed_base_mul(G1W, sw)
    assert _pt1(pts[5]) is None
    w = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32)
    w[:, 7] &= 0x0FFFFFFF
    got = prover.msm_bn254_arrays(pts, w)
    si = [int(a[0]) | int(a[2]) << 128 for a in sw]
    wi = [sum(int(w[i, k]) << (32 * k) for k in range(8)) for i in range(n)]
    assert got == B.mul((1, 2), sum(a * b for a, b in zip(si, wi)) % B.R)
