"""Groth16 over BN254 for the final wrap (GenFinalProof, proto/prover/v1/prover.proto:130-148).

The circuit is service/wrap_circuit.py: an R1CS that verifies all the hashing of the final STARK's verifier at its queries (1.28 M constraints
at the service's parameters), with ONE public input -- the Poseidon commitment to the roots, indices and leaf elements it vouches for -- so the
output still satisfies `[U256; 1]` (src/settlement/ethereum/mod.rs:474-481).  What is real: the prover -- witness completion and A w, B w, C w on
the host (zp_r1cs_eval), the QAP quotient on the GPU (zp_qap_quotient_bn254), five multi-scalar multiplications on the GPU over key points
resident in HBM (zp_msm_bn254 / _g2) -- and proofs that verify under a pairing check (oracle/groth16_verify.py in tests).  What is NOT available
offline: a ceremony.  The key comes from a LOCAL setup with a published seed (toxic waste known -- test keys; its scalars by
zp_r1cs_key_scalars, its group elements by zp_fixed_base_mul_bn254 / _g2 on the GPU), so a proof made here cannot verify under the key in
the reference's contracts/EigenZkVM.json.  The JSON emitted follows the grammar eigen-zeth parses (src/settlement/ethereum/mod.rs:445-481).

A backend without the GPU kernels (the CPU checker of the tests) may offer `groth16_prove(key, witness_ints, rand)` instead: with the toxic
waste in hand the three proof elements are three scalar multiplications -- the SAME group elements (the tests compare the two provers)."""
from __future__ import annotations

import hashlib
import json

import numpy as np

from . import bn254
from .. import native

R = bn254.R
QAP_COSET = 7   # H is computed on the coset 7 <w> (any g with g^m != 1 gives the same coefficients)
DEFAULT_SEED = "zeth-prover-mi355x local test setup v2"


def _g1_words(p):
    out = np.zeros(16, dtype=np.uint32)
    if p is not None:
        for k in range(8):
            out[k] = (p[0] >> (32 * k)) & 0xFFFFFFFF
            out[8 + k] = (p[1] >> (32 * k)) & 0xFFFFFFFF
    return out


def _g2_words(p):
    out = np.zeros(32, dtype=np.uint32)
    if p is not None:
        for c, v in enumerate((p[0][0], p[0][1], p[1][0], p[1][1])):
            for k in range(8):
                out[8 * c + k] = (v >> (32 * k)) & 0xFFFFFFFF
    return out


class Key:
    """toxic: {tau, alpha, beta, gamma, delta}; u, v, l: u64[n_wires][4]; h: u64[m - 1][4] (the key's SCALARS); vk: the verifying key's points;
    dev: device-resident point arrays of a GPU backend (u1x = [u_j]_1 | alpha_1 | delta_1, v1x likewise with beta_1, v2x in G2, l1, h1)"""

    def __init__(self, circuit_blob, seed=DEFAULT_SEED):
        self.blob = np.ascontiguousarray(circuit_blob, dtype=np.uint64)
        self.n_wires, self.logm, self.n_pub = int(self.blob[1]), int(self.blob[3]), int(self.blob[9])
        self.digest = hashlib.sha256(self.blob.tobytes()).hexdigest()
        h = lambda tag: int(hashlib.sha256((seed + "|" + self.digest + "|" + tag).encode()).hexdigest(), 16) % R or 1
        self.toxic = {k: h(k) for k in ("tau", "alpha", "beta", "gamma", "delta")}
        t = self.toxic
        self.u, self.v, self.l, self.h = native.r1cs_key_scalars(self.blob, t["tau"], t["alpha"], t["beta"], t["gamma"], t["delta"])
        li = native.fr_ints(self.l[:1 + self.n_pub])
        self.vk = {"alpha1": bn254.g1_mul(t["alpha"]), "beta2": bn254.g2_mul(t["beta"]), "gamma2": bn254.g2_mul(t["gamma"]), "delta2": bn254.g2_mul(t["delta"]),
                   "ic": [bn254.g1_mul(x) if x else None for x in li]}
        self.dev = None

    def load_points(self, be):
        """the key's group elements, made on the GPU from its scalars and left in HBM (once per key and backend)"""
        if self.dev is not None:
            return self.dev
        p, t = be.p, self.toxic
        g1, g2 = _g1_words(bn254.G1), _g2_words(bn254.G2)
        tail_a = native.fr_words([t["alpha"], t["delta"]])
        tail_b = native.fr_words([t["beta"], t["delta"]])
        l_priv = self.l.copy()
        l_priv[:1 + self.n_pub] = 0                       # the public part of C is the verifier's (IC), not the prover's
        dev = {}
        for name, sc, is_g2 in (("u1x", np.concatenate([self.u, tail_a]), False), ("v1x", np.concatenate([self.v, tail_b]), False),
                                ("v2x", np.concatenate([self.v, tail_b]), True), ("l1", l_priv, False), ("h1", self.h, False)):
            pts = p.fixed_base_mul(g2 if is_g2 else g1, sc, g2=is_g2)
            d = p.alloc(pts.size // 2)
            p._chk(p.lib.zp_h2d(p.ctx, d.ptr, pts.ctypes.data, pts.nbytes))
            dev[name] = (d, pts.shape[0])
        dev["delta1"] = bn254.g1_mul(t["delta"])
        self.dev = dev
        return dev


def prove(key, witness, a_ev, b_ev, c_ev, be, rand):
    """witness u64[n_wires][4] (complete), a_ev / b_ev / c_ev u64[2^logm][4] from zp_r1cs_eval; rand = (r, s).
    Returns ({"pi_a", "pi_b", "pi_c"}, [public inputs as ints])."""
    pub = native.fr_ints(witness[1:1 + key.n_pub])
    if hasattr(be, "groth16_prove"):                 # a backend that proves another way (the CPU checker: by the trapdoor)
        return be.groth16_prove(key, native.fr_ints(witness), rand), pub
    p = be.p
    dev = key.load_points(be)
    n, m = key.n_wires, 1 << key.logm
    r, s = rand
    # H = (A B - C) / Z on the GPU, its coefficients stay in HBM and are the scalars of the h MSM
    d_a, d_b, d_c = p.upload(a_ev.reshape(-1)), p.upload(b_ev.reshape(-1)), p.upload(c_ev.reshape(-1))
    cw = native.fr_words([QAP_COSET])
    p._chk(p.lib.zp_qap_quotient_bn254(p.ctx, d_a.ptr, d_b.ptr, d_c.ptr, key.logm, cw.ctypes.data))
    # the witness as MSM scalars, two extra entries for the blinding terms that ride in the MSMs: [w | 1 | r or s]
    sc = np.concatenate([witness, native.fr_words([1, r])])
    d_s = p.upload(sc.reshape(-1))

    def msm(points, count, g2=False):
        import ctypes as C
        out = (C.c_uint32 * (32 if g2 else 16))()
        fn = p.lib.zp_msm_bn254_g2 if g2 else p.lib.zp_msm_bn254
        p._chk(fn(p.ctx, points[0].ptr, d_s.ptr, count, out))
        if g2:
            v = [sum(int(out[8 * c + k]) << (32 * k) for k in range(8)) for c in range(4)]
            return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))
        x = sum(int(out[k]) << (32 * k) for k in range(8))
        y = sum(int(out[8 + k]) << (32 * k) for k in range(8))
        return None if (x == 0 and y == 0) else (x, y)
    A1 = msm(dev["u1x"], n + 2)                                          # alpha + sum_j w_j u_j + r delta
    last = native.fr_words([s])
    p._chk(p.lib.zp_h2d(p.ctx, d_s.offset(4 * (n + 1)), last.ctypes.data, 32))
    B1 = msm(dev["v1x"], n + 2)                                          # beta + sum_j w_j v_j + s delta
    B2 = msm(dev["v2x"], n + 2, g2=True)
    Cl = msm(dev["l1"], n)                                               # the key's l points of the public wires are infinity
    import ctypes as C
    out = (C.c_uint32 * 16)()
    p._chk(p.lib.zp_msm_bn254(p.ctx, dev["h1"][0].ptr, d_a.ptr, m - 1, out))      # sum_i H_i [tau^i Z(tau) / delta]
    hx = sum(int(out[k]) << (32 * k) for k in range(8))
    hy = sum(int(out[8 + k]) << (32 * k) for k in range(8))
    Ch = None if (hx == 0 and hy == 0) else (hx, hy)
    tail = be.msm_g1([A1, B1, dev["delta1"]], [s, r, (R - r * s % R) % R])
    Cp = bn254._pt_add(bn254._Ops1, bn254._pt_add(bn254._Ops1, Cl, Ch), tail)
    for d in (d_a, d_b, d_c, d_s):
        d.free()
    return {"pi_a": A1, "pi_b": B2, "pi_c": Cp}, pub


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
