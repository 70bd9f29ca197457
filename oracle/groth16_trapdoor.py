"""oracle/groth16_trapdoor.py -- the checker's Groth16 prover (TEST INFRASTRUCTURE; imports nothing from the product package).

The product's keys come from a local, seeded setup (the toxic waste tau, alpha, beta, gamma, delta is known: test keys, not a ceremony).
With it in hand the three elements of a Groth16 proof are three scalar multiplications:
    a = alpha + sum_j w_j u_j(tau) + r delta                                   A  = a G1
    b = beta  + sum_j w_j v_j(tau) + s delta                                   B  = b G2
    c = sum_{private j} w_j l_j + H(tau) Z(tau) / delta + s a + r b - r s delta    C  = c G1
with H(tau) Z(tau) = (sum_j w_j u_j)(sum_j w_j v_j) - sum_j w_j w_j(tau) and w_j(tau) recovered from l_j = (beta u_j + alpha v_j + w_j) / delta (or
/ gamma for the constant and the public wires).  These are the SAME group elements an honest prover obtains from the key's points with a
QAP quotient and five multi-scalar multiplications -- so a proof from the GPU (2^21-point transforms and MSMs) can be compared with this one
byte for byte (deterministic blinding), which checks those kernels at the size the wrap uses.  PARITY UNPINNED w.r.t. the external prover."""
from . import naive_bn254 as B

R = B.R


def prove(toxic, u, v, l, n_pub, witness, rand):
    """toxic: dict of ints; u, v, l: per-wire scalars (lists of ints); witness: list of ints (wire 0 = 1); rand = (r, s)"""
    al, be, ga, de = toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"]
    r, s = rand
    U = sum(w * x for w, x in zip(witness, u)) % R
    V = sum(w * x for w, x in zip(witness, v)) % R
    Wc = 0
    for j, (w, lj, uj, vj) in enumerate(zip(witness, l, u, v)):
        wj = (lj * (ga if j <= n_pub else de) - be * uj - al * vj) % R
        Wc = (Wc + w * wj) % R
    a = (al + U + r * de) % R
    b = (be + V + s * de) % R
    priv = sum(w * lj for j, (w, lj) in enumerate(zip(witness, l)) if j > n_pub) % R
    c = (priv + (U * V - Wc) * pow(de, -1, R) + s * a + r * b - r * s % R * de) % R
    return {"pi_a": B.mul((1, 2), a), "pi_b": B.mul_g2(B.G2, b), "pi_c": B.mul((1, 2), c)}
