"""Groth16 over BN254 for the final wrap (GenFinalProof, proto/prover/v1/prover.proto:130-148).

What is real here: a complete Groth16 prover -- R1CS -> QAP, CRS, proof (A, B, C) with the G1
multi-scalar multiplications on the GPU (zp_msm_bn254) -- whose proofs verify under a pairing check
(oracle/groth16_verify.py in tests).  What is NOT available offline: the circuit that verifies the
recursive STARK (circom/R1CS of eigen-zkvm) and its ceremony CRS.  The circuit proven instead is a
fixed arithmetic chain binding the public input to a secret derived from the aggregated proof:
    x_0 = s,  x_(i+1) = x_i * s + (i+1)  (i < steps),  pub = x_steps + s
(the right-hand factor of every product is the single wire s, so the G2 side of B is one scalar multiplication)
and the CRS comes from a LOCAL setup with a published seed (toxic waste known -- test setup, not a
ceremony).  The JSON emitted follows the grammar eigen-zeth parses (src/settlement/ethereum/mod.rs:445-481).
"""
from __future__ import annotations

import hashlib
import json
import os

from . import bn254

R = bn254.R
TWO_ADIC_ROOT = pow(5, (R - 1) >> 28, R)   # 5 generates F_r^*; 2-adicity of r-1 is 28


def _root(logm):
    return pow(TWO_ADIC_ROOT, 1 << (28 - logm), R)


class Circuit:
    """R1CS of the arithmetic chain.  wires: [1, pub, s, x_1 .. x_steps]"""

    def __init__(self, logm=8):
        self.logm, self.m = logm, 1 << logm
        self.steps = self.m - 2
        self.nwires = 3 + self.steps
        self.npub = 1
        A, B, C = [], [], []
        prev = 2
        for i in range(self.steps):
            nxt = 3 + i
            A.append({prev: 1}); B.append({2: 1}); C.append({nxt: 1, 0: (-(i + 1)) % R})
            prev = nxt
        A.append({prev: 1, 2: 1}); B.append({0: 1}); C.append({1: 1})
        self.A, self.B, self.C = A, B, C

    def witness(self, s):
        s %= R
        w = [1, 0, s]
        x = s
        for i in range(self.steps):
            x = (x * s + (i + 1)) % R
            w.append(x)
        w[1] = (x + s) % R
        return w

    def check(self, w):
        dot = lambda row: sum(c * w[j] for j, c in row.items()) % R
        return all(dot(a) * dot(b) % R == dot(c) for a, b, c in zip(self.A, self.B, self.C))


def setup(circ, seed="zeth-prover-mi355x local test setup v1"):
    """CRS from a published seed (NOT a ceremony).  Returns (proving key, verifying key)."""
    h = lambda tag: int(hashlib.sha256((seed + "|" + tag).encode()).hexdigest(), 16) % R or 1
    tau, alpha, beta, gamma, delta = h("tau"), h("alpha"), h("beta"), h("gamma"), h("delta")
    m, w = circ.m, _root(circ.logm)
    # Lagrange basis at tau: L_i(tau) = (tau^m - 1) w^i / (m (tau - w^i))
    zt = (pow(tau, m, R) - 1) % R
    minv = pow(m, R - 2, R)
    lag = [zt * pow(w, i, R) % R * minv % R * pow((tau - pow(w, i, R)) % R, R - 2, R) % R for i in range(m)]
    u, v, ww = [0] * circ.nwires, [0] * circ.nwires, [0] * circ.nwires
    for i, (a, b, c) in enumerate(zip(circ.A, circ.B, circ.C)):
        for j, cf in a.items(): u[j] = (u[j] + cf * lag[i]) % R
        for j, cf in b.items(): v[j] = (v[j] + cf * lag[i]) % R
        for j, cf in c.items(): ww[j] = (ww[j] + cf * lag[i]) % R
    ginv, dinv = pow(gamma, R - 2, R), pow(delta, R - 2, R)
    g1, g2 = bn254.g1_mul, bn254.g2_mul
    pk = {
        "alpha1": g1(alpha), "beta1": g1(beta), "beta2": g2(beta), "delta1": g1(delta), "delta2": g2(delta),
        "u1": [g1(x) if x else (0, 0) for x in u],
        "v1": [g1(x) if x else (0, 0) for x in v],
        "v2": [g2(x) if x else None for x in v],
        "l1": [None] * (1 + circ.npub) + [g1((beta * u[j] + alpha * v[j] + ww[j]) % R * dinv % R)
                                          for j in range(1 + circ.npub, circ.nwires)],
        "h1": [g1(pow(tau, i, R) * zt % R * dinv % R) for i in range(m - 1)],
    }
    vk = {"alpha1": pk["alpha1"], "beta2": pk["beta2"], "gamma2": g2(gamma), "delta2": pk["delta2"],
          "ic": [g1((beta * u[j] + alpha * v[j] + ww[j]) % R * ginv % R) for j in range(1 + circ.npub)]}
    return pk, vk


def _neg(p):
    return None if p is None else (p[0], (-p[1]) % bn254.P)


def _g1_add(a, b):
    return bn254._pt_add(bn254._Ops1, a, b)


def _g2_add(a, b):
    return bn254._pt_add(bn254._Ops2, a, b)


QAP_COSET = 7   # H is computed on the coset 7 <w> (any g with g^m != 1 gives the same coefficients)


def prove(circ, pk, w, msm_g1, rand, msm_g2, qap_quotient):
    """msm_g1(points[(x,y)], scalars[int]) -> (x, y) | None   (GPU: Prover.msm_bn254);  rand = (r, s);
    msm_g2(points[((x0,x1),(y0,y1)) | None], scalars) -> point | None  (GPU: Prover.msm_bn254_g2) for the G2 side of B;
    qap_quotient(a_ev, b_ev, c_ev, logm, coset) -> coefficients of H = (A B - C) / (x^m - 1)  (GPU: zp_qap_quotient_bn254).
    msm_g2 may be None (a backend without a G2 kernel: the CPU checker); the other two are required -- no host fallback."""
    assert circ.check(w)
    m = circ.m
    dot = lambda row: sum(c * w[j] for j, c in row.items()) % R
    a_ev = [dot(r) for r in circ.A] + [0] * (m - len(circ.A))
    b_ev = [dot(r) for r in circ.B] + [0] * (m - len(circ.B))
    c_ev = [dot(r) for r in circ.C] + [0] * (m - len(circ.C))
    g = QAP_COSET
    hco = qap_quotient(a_ev, b_ev, c_ev, circ.logm, g)
    assert len(hco) == m and hco[m - 1] == 0
    r, s = rand
    # the blinding terms ride in the multi-scalar multiplications (alpha, beta, delta as extra bases with scalars 1, r, s):
    # no scalar multiplication is left to host code
    A1 = msm_g1(pk["u1"] + [pk["alpha1"], pk["delta1"]], list(w) + [1, r])
    B1 = msm_g1(pk["v1"] + [pk["beta1"], pk["delta1"]], list(w) + [1, s])
    if msm_g2 is not None:                       # beta + sum_j w_j [v_j(tau)]_2 + s delta as one G2 multi-scalar multiplication
        B2 = msm_g2(pk["v2"] + [pk["beta2"], pk["delta2"]], list(w) + [1, s])
    else:
        B2 = pk["beta2"]
        for j, pt in enumerate(pk["v2"]):
            if pt is not None and w[j]:
                B2 = _g2_add(B2, bn254._pt_mul(bn254._Ops2, pt, w[j]))
        B2 = _g2_add(B2, bn254._pt_mul(bn254._Ops2, pk["delta2"], s))
    priv = list(range(1 + circ.npub, circ.nwires))
    Cp = msm_g1([pk["l1"][j] for j in priv] + pk["h1"] + [A1, B1, pk["delta1"]],
                [w[j] for j in priv] + hco[:m - 1] + [s, r, (R - r * s % R) % R])
    return {"pi_a": A1, "pi_b": B2, "pi_c": Cp}, [w[1]]


def proof_to_json(proof, extra=None):
    a, b, c = proof["pi_a"], proof["pi_b"], proof["pi_c"]
    d = {"pi_a": {"x": str(a[0]), "y": str(a[1])},
         "pi_b": {"x": [str(b[0][0]), str(b[0][1])], "y": [str(b[1][0]), str(b[1][1])]},
         "pi_c": {"x": str(c[0]), "y": str(c[1])}, "protocol": "groth16", "curve": "BN128"}
    d.update(extra or {})
    return json.dumps(d)


def vk_to_json(vk):
    f2 = lambda p: {"x": [str(p[0][0]), str(p[0][1])], "y": [str(p[1][0]), str(p[1][1])]}
    f1 = lambda p: {"x": str(p[0]), "y": str(p[1])}
    return json.dumps({"alpha1": f1(vk["alpha1"]), "beta2": f2(vk["beta2"]), "gamma2": f2(vk["gamma2"]),
                       "delta2": f2(vk["delta2"]), "ic": [f1(p) for p in vk["ic"]]})


def _enc(o):
    """CRS as plain JSON: ints as decimal strings, points as lists, None (infinity) as null -- nothing executable"""
    if isinstance(o, dict):
        return {k: _enc(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_enc(v) for v in o]
    if isinstance(o, int):
        return str(o)
    return o


def _dec(o):
    if isinstance(o, dict):
        return {k: _dec(v) for k, v in o.items()}
    if isinstance(o, list):
        vals = [_dec(v) for v in o]
        # points are tuples in the prover: (x, y) over F_q, ((x0, x1), (y0, y1)) over F_q2
        if len(vals) == 2 and all(isinstance(v, (int, tuple)) for v in vals):
            return tuple(vals)
        return vals
    if isinstance(o, str):
        return int(o)
    return o


def load_or_setup(circ, cache_dir):
    """The CRS of the stand-in circuit is deterministic (SEEDED: a test-only key, its toxic waste is public) and cached
    as JSON in a directory only this user can write (0700); the file is written through a unique temporary name."""
    import tempfile
    os.makedirs(cache_dir, mode=0o700, exist_ok=True)
    st = os.stat(cache_dir)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise PermissionError("CRS cache directory %s must be owned by this user and not writable by others" % cache_dir)
    path = os.path.join(cache_dir, "groth16_crs_v3_logm%d.json" % circ.logm)
    if os.path.exists(path):
        with open(path) as f:
            d = _dec(json.load(f))
        return d["pk"], d["vk"]
    pk, vk = setup(circ)
    fd, tmp = tempfile.mkstemp(prefix=".crs-", dir=cache_dir)
    with os.fdopen(fd, "w") as f:
        json.dump({"pk": _enc(pk), "vk": _enc(vk)}, f)
    os.replace(tmp, path)
    return pk, vk
