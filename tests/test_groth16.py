"""Final-wrap tests: the pure-Python pairing (oracle) is sane, the Groth16 prover (product) emits proofs that
satisfy the pairing equation, tampering is rejected; on the GPU the G1 MSMs run through zp_msm_bn254 and the
proof equals the CPU one."""
import pytest

from eigen_zeth_amd.service import bn254
from eigen_zeth_amd.service import groth16 as G
from oracle import bn254_pairing as BP
from oracle import groth16_verify as GV
from oracle import naive_bn254 as B1
from oracle import naive as NV

QAP = lambda a, b, c, logm, g: NV.qap_quotient(a, b, c) if logm <= 5 else None   # definition-level H (the checker)


@pytest.fixture(scope="module")
def keys():
    c = G.Circuit(4)
    pk, vk = G.setup(c)
    return c, pk, vk


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


def test_groth16_verifies_and_rejects(keys):
    c, pk, vk = keys
    w = c.witness(987654321)
    assert c.check(w)
    proof, pub = G.prove(c, pk, w, B1.msm, (5, 9), None, QAP)
    assert GV.verify(vk, proof, pub)
    assert not GV.verify(vk, proof, [(pub[0] + 1) % G.R])
    bad = dict(proof)
    bad["pi_a"] = B1.add(proof["pi_a"], B1.G)
    assert not GV.verify(vk, bad, pub)
    wbad = list(w)
    wbad[5] = (wbad[5] + 1) % G.R
    with pytest.raises(AssertionError):
        G.prove(c, pk, wbad, B1.msm, (5, 9), None, QAP)   # an unsatisfying witness is refused
    # the JSON form is the grammar eigen-zeth parses
    import json
    js = json.loads(G.proof_to_json(proof))
    assert js["protocol"] == "groth16" and js["curve"] == "BN128" and js["pi_b"]["x"][0].isdigit()


@pytest.mark.gpu
def test_groth16_with_gpu_msm_equals_cpu(prover, keys):
    c, pk, vk = keys
    w = c.witness(55555)
    gpu_msm = lambda pts, sc: prover.msm_bn254([p if p is not None else (0, 0) for p in pts], [int(s) for s in sc])
    pg, pubg = G.prove(c, pk, w, gpu_msm, (7, 8), None, lambda a, b, cc, logm, g: prover.qap_quotient_bn254(a, b, cc, logm, g))
    pc, pubc = G.prove(c, pk, w, B1.msm, (7, 8), None, QAP)
    assert pg == pc and pubg == pubc
    assert GV.verify(vk, pg, pubg)
