"""oracle/groth16_verify.py -- Groth16 verification equation (TEST INFRASTRUCTURE):
e(A,B) = e(alpha,beta) * e(sum_j pub_j IC_j, gamma) * e(C, delta), checked as one product of Miller loops."""
from . import bn254_pairing as BP
from . import naive_bn254 as B1


def verify(vk, proof, publics):
    """vk: dict alpha1, beta2, gamma2, delta2, ic[];  proof: pi_a (G1), pi_b (G2), pi_c (G1) as int tuples"""
    if len(publics) + 1 != len(vk["ic"]):
        return False
    if not (B1.on_curve(proof["pi_a"]) and B1.on_curve(proof["pi_c"])):
        return False
    vkx = vk["ic"][0]
    for pub, ic in zip(publics, vk["ic"][1:]):
        vkx = B1.add(vkx, B1.mul(ic, pub % B1.R))
    neg = lambda p: (p[0], (-p[1]) % B1.Q)
    f = BP.miller(proof["pi_b"], proof["pi_a"])
    f = BP.f_mul(f, BP.miller(vk["beta2"], neg(vk["alpha1"])))
    f = BP.f_mul(f, BP.miller(vk["gamma2"], neg(vkx)))
    f = BP.f_mul(f, BP.miller(vk["delta2"], neg(proof["pi_c"])))
    return BP.final_exp(f) == BP.ONE
