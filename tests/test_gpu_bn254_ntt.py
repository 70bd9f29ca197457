"""GPU parity for the Groth16 QAP step (zp_ntt_bn254, zp_qap_quotient_bn254) through the C-ABI: golden vectors
(definition level), the recursive checker at sizes with 2 and 3 passes, direct evaluations, round trips, linearity."""
import json
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(HERE, "golden", "vectors.json")) as f:
        return json.load(f)


def _run(p, x, logn, inverse=False, coset=None):
    d = p.upload(p._fr_words(x).reshape(-1))
    p.ntt_bn254(d, logn, inverse, coset)
    out = p._fr_ints(p.download(d, (len(x), 4)))
    d.free()
    return out


def test_fr_ntt_golden(prover, golden):
    for case in golden["fr_ntt"]:
        x, g = [int(v) for v in case["in"]], int(case["coset"])
        logn = len(x).bit_length() - 1
        assert _run(prover, x, logn, False, g if g != 1 else None) == [int(v) for v in case["forward"]]
        assert _run(prover, x, logn, True, g if g != 1 else None) == [int(v) for v in case["inverse"]]
        assert _run(prover, x, logn, False, g) == [int(v) for v in case["forward"]]     # g = 1 given explicitly


@pytest.mark.parametrize("logn", [3, 5, 8, 9, 11, 13, 16, 17])
def test_fr_ntt_matches_checker(prover, logn):
    from oracle import naive as NV
    rnd = random.Random(100 + logn)
    n = 1 << logn
    x = [rnd.randrange(NV.FR) for _ in range(n)]
    x[0], x[n - 1] = NV.FR - 1, 0
    g = rnd.randrange(2, NV.FR)
    assert _run(prover, x, logn) == NV.fr_ntt_fast(x)
    y = _run(prover, x, logn, False, g)
    assert y == NV.fr_ntt_fast(x, coset=g)
    assert _run(prover, y, logn, True, g) == x
    assert _run(prover, x, logn, True) == NV.fr_ntt_fast(x, inverse=True)


def test_fr_ntt_large_properties(prover):
    """2^20 points (three passes 7 + 7 + 6): round trip, direct evaluation of outputs of a sparse input, linearity"""
    from oracle import naive as NV
    logn, n, g = 20, 1 << 20, 7
    probes = [0, 1, 12345, n - 1]
    rng = np.random.default_rng(5)
    words = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    words[:, 3] &= np.uint64((1 << 60) - 1)                                 # every value < 2^252 < r
    pos = sorted(set(int(v) for v in rng.integers(0, n, size=64)))
    sparse, rest = np.zeros_like(words), words.copy()
    for i in pos:
        sparse[i], rest[i] = words[i], 0

    def fwd(arr):
        d = prover.upload(arr.reshape(-1))
        prover.ntt_bn254(d, logn, False, g)
        y = prover.download(d, (n, 4))
        prover.ntt_bn254(d, logn, True, g)
        back = prover.download(d, (n, 4))
        d.free()
        assert (back == arr).all()                                          # round trip
        return y

    y_all, y_sparse, y_rest = fwd(words), fwd(sparse), fwd(rest)
    w, x = NV.fr_root(logn), prover._fr_ints(words)
    for k in probes:                                                        # y_k = sum_i x_i (g w^k)^i, from the definition
        z = g * pow(w, k, NV.FR) % NV.FR
        assert prover._fr_ints(y_sparse[k])[0] == sum(x[i] * pow(z, i, NV.FR) for i in pos) % NV.FR
    a, b, c = (prover._fr_ints(v[probes]) for v in (y_all, y_sparse, y_rest))
    assert [(u + v) % NV.FR for u, v in zip(b, c)] == a                     # linearity


def test_qap_quotient_golden(prover, golden):
    for case in golden["qap_quotient"]:
        a, b, c, h = ([int(v) for v in case[k]] for k in "abch")
        logm = len(a).bit_length() - 1
        for g in (7, 5, 0x1234567890abcdef):
            assert prover.qap_quotient_bn254(a, b, c, logm, g) == h


def test_qap_quotient_identity_2_12(prover):
    from oracle import naive as NV
    rnd = random.Random(9)
    logm, m = 12, 1 << 12
    a = [rnd.randrange(NV.FR) for _ in range(m)]
    b = [rnd.randrange(NV.FR) for _ in range(m)]
    c = [a[i] * b[i] % NV.FR for i in range(m)]
    h = prover.qap_quotient_bn254(a, b, c, logm, 7)
    assert h[-1] == 0
    A, B, C = (NV.fr_ntt_fast(v, inverse=True) for v in (a, b, c))
    z = rnd.randrange(NV.FR)
    ev = lambda co: NV.poly_eval_mod(co, z, NV.FR)
    assert (ev(A) * ev(B) - ev(C)) % NV.FR == ev(h) * (pow(z, m, NV.FR) - 1) % NV.FR


def test_bad_arguments(prover):
    from eigen_zeth_amd.native import ZpError
    d = prover.alloc(4 * 16)
    with pytest.raises(ZpError):
        prover.ntt_bn254(d, 29)
    with pytest.raises(ZpError):
        prover.ntt_bn254(d, 4, False, 0)
    with pytest.raises(ZpError):      # the coset shift 1 lies in the domain: Z vanishes
        prover.qap_quotient_bn254([1] * 4, [1] * 4, [1] * 4, 2, 1)
    d.free()
