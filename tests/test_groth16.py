"""The checker's pairing and the product's host curve helpers (the Groth16 prover itself: tests/test_wrap_circuit.py on the CPU,
tests/test_gpu_wrap.py on the MI355X)."""
import pytest

from eigen_zeth_amd.service import bn254
from oracle import bn254_pairing as BP
from oracle import naive_bn254 as B1


def test_pairing_bilinear_nondegenerate():
    e1 = BP.pairing(bn254.G2, bn254.G1)
    assert e1 != BP.ONE and BP.f_pow(e1, BP.R) == BP.ONE
    assert BP.pairing(bn254.g2_mul(6), bn254.g1_mul(35)) == BP.f_pow(e1, 210)
    assert BP.f_mul(BP.f_inv(e1), e1) == BP.ONE


def test_host_curve_helpers_match_definition():
    for k in (1, 2, 3, 12345678901234567890, bn254.R - 1):
        assert bn254.g1_mul(k) == B1.mul(B1.G, k)
        assert bn254.g2_on_curve(bn254.g2_mul(k))
    assert bn254.g1_mul(bn254.R) is None
