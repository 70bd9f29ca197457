"""The verifying key embedded in the reference's verifier contract (contracts/EigenZkVM.json) as a committed fixture:
tests/golden/ref_vk.json, made by tests/golden/extract_ref_vk.py.  These 18 constants and the
proof fixtures are the only BN254 data the reference holds; here they pin the checker's curve / twist / pairing code on
reference-held data, and record as a committed NEGATIVE result that the reference's own proof fixture
(proof/proof.json + proof/public_input.json = tests/golden/ref_proof.json, ref_public_input.json) does not satisfy the
Groth16 equation under this key for ANY assignment of the pushed points to the roles alpha / beta / gamma / delta / IC
(the fixtures pin the JSON shape the settlement parser accepts, src/settlement/ethereum/mod.rs:445-481,
not any arithmetic).  On-chain check this key belongs to: src/settlement/ethereum/interfaces/zkvm.rs:82-130."""
import itertools
import json
import os

import pytest

from oracle import bn254_pairing as BP
from oracle import naive_bn254 as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/contracts/EigenZkVM.json"


@pytest.fixture(scope="module")
def vk():
    with open(os.path.join(ROOT, "tests", "golden", "ref_vk.json")) as f:
        d = json.load(f)
    g1 = [tuple(int(v) for v in p) for p in d["g1"]]
    g2 = [((int(w[0]), int(w[1])), (int(w[2]), int(w[3]))) for w in d["g2_words"]]
    return d, g1, g2


def test_fixture_is_what_the_extractor_produces(vk):
    """where the reference is present the committed fixture is regenerated and compared (it is absent on the GPU box)"""
    if not os.path.exists(REF):
        pytest.skip("reference not present on this machine")
    import importlib.util
    spec = importlib.util.spec_from_file_location("extract_ref_vk", os.path.join(ROOT, "tests", "golden", "extract_ref_vk.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    code = bytes.fromhex(json.load(open(REF))["deployedBytecode"]["object"][2:])
    got = [(o, n, str(v)) for (o, n, v) in ex.pushes(code) if n >= 24 and ex.WINDOW[0] <= o <= ex.WINDOW[1]]
    assert got == [(c["offset"], c["push_bytes"], c["value"]) for c in vk[0]["constants"]]
    assert len(code) == vk[0]["bytecode_bytes"]


def test_layout_of_the_pushes(vk):
    d = vk[0]
    offs = [c["offset"] for c in d["constants"]]
    assert len(offs) == 18 and offs == sorted(offs) and offs[0] == 10542 and offs[-1] == 11316
    assert [c["push_bytes"] for c in d["constants"]].count(31) == 1          # one coordinate has a leading zero byte (PUSH31)
    assert all(int(c["value"]) < B.Q for c in d["constants"])                 # every constant is a base-field element
    assert sorted(d["moduli_pushes"].values()) == ["P", "R"]                  # the same bytecode pushes both BN254 moduli


def test_g1_points_are_on_the_curve(vk):
    _, g1, _ = vk
    assert all(B.on_curve(p) for p in g1)
    # the third point is pushed y-before-x: in push order it is NOT on the curve
    c = [int(x["value"]) for x in vk[0]["constants"]]
    assert not B.on_curve((c[16], c[17])) and B.on_curve((c[17], c[16]))
    assert len(set(g1)) == 3


def test_g2_points_are_on_the_twist_and_in_the_subgroup(vk):
    """each group of four words is a point of E'(F_q2): y^2 = x^3 + 3/(9+u) with x = (w0, w1), y = (w2, w3) -- and only in
    that order -- and has order r (G2 is a proper subgroup of the twist, so this is not implied by being on it)"""
    _, _, g2 = vk
    for p in g2:
        assert B.on_curve_g2(p)
        assert not B.on_curve_g2(((p[0][1], p[0][0]), (p[1][1], p[1][0])))
        assert B.mul_g2(p, B.R) is None
    assert len(set(g2)) == 3


def test_pairing_is_bilinear_on_the_reference_points(vk):
    """e(2 P, Q) = e(P, Q)^2 = e(P, 2 Q) with P, Q taken from the reference's key (the pairing code's first reference-held input)"""
    _, g1, g2 = vk
    P, Q2 = g1[0], g2[0]
    e = BP.pairing(Q2, P)
    assert e != BP.ONE
    assert BP.pairing(Q2, B.mul(P, 2)) == BP.f_mul(e, e)
    assert BP.pairing(B.mul_g2(Q2, 2), P) == BP.f_mul(e, e)


def test_reference_proof_fixture_verifies_under_no_role_assignment(vk):
    """Groth16: e(A, B) = e(alpha, beta) e(IC0 + pub IC1, gamma) e(C, delta).  The pushed constants do not say which G2 point
    is beta / gamma / delta, nor which G1 point is alpha / IC0 / IC1: all 6 x 6 assignments are tried, each as one product
    of Miller loops and one final exponentiation.  None holds -- the fixture is a format fixture."""
    _, g1, g2 = vk
    pr = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_proof.json")))
    pub = int(json.load(open(os.path.join(ROOT, "tests", "golden", "ref_public_input.json")))[0])
    A = (int(pr["pi_a"]["x"]), int(pr["pi_a"]["y"]))
    Bp = ((int(pr["pi_b"]["x"][0]), int(pr["pi_b"]["x"][1])), (int(pr["pi_b"]["y"][0]), int(pr["pi_b"]["y"][1])))
    Cp = (int(pr["pi_c"]["x"]), int(pr["pi_c"]["y"]))
    assert B.on_curve(A) and B.on_curve(Cp) and B.on_curve_g2(Bp) and pub < B.R
    neg = lambda p: (p[0], (-p[1]) % B.Q)
    m_ab = BP.miller(Bp, A)
    m_alpha = {(i, j): BP.miller(g2[j], neg(g1[i])) for i in range(3) for j in range(3)}            # e(-alpha, beta)
    m_c = {j: BP.miller(g2[j], neg(Cp)) for j in range(3)}                                          # e(-C, delta)
    m_x = {}
    for i0, i1 in itertools.permutations(range(3), 2):
        vkx = B.add(g1[i0], B.mul(g1[i1], pub))
        for j in range(3):
            m_x[(i0, i1, j)] = BP.miller(g2[j], neg(vkx))                                            # e(-vk_x, gamma)
    accepted = []
    for (ia, i0, i1) in itertools.permutations(range(3), 3):
        for (jb, jg, jd) in itertools.permutations(range(3), 3):
            f = BP.f_mul(BP.f_mul(m_ab, m_alpha[(ia, jb)]), BP.f_mul(m_x[(i0, i1, jg)], m_c[jd]))
            if BP.final_exp(f) == BP.ONE:
                accepted.append((ia, i0, i1, jb, jg, jd))
    assert accepted == []
