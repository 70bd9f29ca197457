"""Groth16 over BN254 for the final wrap (GenFinalProof, proto/prover/v1/prover.proto:130-148).

The circuit is service/wrap_circuit.py: an R1CS that verifies all the hashing of the final STARK's verifier at its queries (1.28 M constraints
at the service's parameters), with ONE public input -- the Poseidon commitment to the roots, indices and leaf elements it vouches for -- so the
output still satisfies `[U256; 1]` (src/settlement/ethereum/mod.rs:474-481).  What is real: the prover -- ONE library call (zp_groth16_prove: witness completion and A w, B w, C w,
the QAP quotient on the GPU, five multi-scalar multiplications on the GPU over key points resident in HBM, the blinding terms) -- and proofs that verify under a pairing check (oracle/groth16_verify.py in tests).  What is NOT available
offline: a ceremony.  The key comes from a LOCAL setup with a published seed (toxic waste known -- test keys; its scalars by
zp_r1cs_key_scalars, its group elements by zp_fixed_base_mul_bn254 / _g2 on the GPU), so a proof made here cannot verify under the key in
the reference's contracts/EigenZkVM.json.  The JSON emitted follows the grammar eigen-zeth parses (src/settlement/ethereum/mod.rs:445-481).

The prover is a method of the backend, like every other stage (`be.groth16(key, set_idx, set_val, rand)`; HipBackend: prove_on_gpu below).  The
tests' CPU backend implements it its own way -- with the toxic waste of the test key in hand the three proof elements are three scalar
multiplications, the SAME group elements (tests/cpu_wrap_backend.py; the tests compare the two provers)."""
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
    dev: device-resident point arrays of a GPU backend (u1x = [u_j]_1 | alpha_1 | delta_1; v_wires: the wires with a non-zero column in B, v1x =
    [v_j]_1 of those | beta_1 | delta_1, v2x the same in G2; l1, h1)"""

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
        # a third of the gadget's wires (the x^4 of every S-box) never stand in B: the B side of the key holds the other wires only
        v_wires = np.flatnonzero((self.v != 0).any(axis=1)).astype(np.uint32)
        vs = np.concatenate([self.v[v_wires], tail_b])
        for name, sc, is_g2 in (("u1x", np.concatenate([self.u, tail_a]), False), ("v1x", vs, False), ("v2x", vs, True), ("l1", l_priv, False),
                                ("h1", self.h, False)):
            pts = p.fixed_base_mul(g2 if is_g2 else g1, sc, g2=is_g2)
            d = p.alloc(pts.size // 2)
            p._chk(p.lib.zp_h2d(p.ctx, d.ptr, pts.ctypes.data, pts.nbytes))
            dev[name] = (d, pts.shape[0])
        d = p.alloc((v_wires.size + 1) // 2)
        p._chk(p.lib.zp_h2d(p.ctx, d.ptr, v_wires.ctypes.data, v_wires.nbytes))
        dev["v_wires"] = (d, int(v_wires.size))
        dev["delta1"] = bn254.g1_mul(t["delta"])
        self.dev = dev
        return dev


def _g1_point(w):
    x = sum(int(w[k]) << (32 * k) for k in range(8))
    y = sum(int(w[8 + k]) << (32 * k) for k in range(8))
    return None if (x == 0 and y == 0) else (x, y)


def _g2_point(w):
    v = [sum(int(w[8 * c + k]) << (32 * k) for k in range(8)) for c in range(4)]
    return None if not any(v) else ((v[0], v[1]), (v[2], v[3]))


def prove(key, set_idx, set_val, be, rand):
    """set_idx u64[n], set_val u64[n][4]: the caller-set wires (zp_wrap_assign, or WrapCircuit.assign's witness where its mask is set);
    rand = (r, s).  Returns ({"pi_a", "pi_b", "pi_c"}, [public inputs as ints], [ms witness, ms QAP, ms MSMs]); ValueError when the assignment
    does not satisfy the circuit.  The backend's stage (HipBackend.groth16 -> prove_on_gpu: ONE library call, zp_groth16_prove)."""
    return be.groth16(key, set_idx, set_val, rand)


def prove_on_gpu(key, set_idx, set_val, be, rand):
    """zp_groth16_prove on the backend's ctx over the key points resident in HBM"""
    dev = key.load_points(be)
    handles = {k: v[0] for k, v in dev.items() if k != "delta1"}
    handles["n_v"] = dev["v_wires"][1]
    a, b, c, pub, ms = be.p.groth16_prove(key.blob, handles, _g1_words(dev["delta1"]), set_idx, set_val, *rand)
    return {"pi_a": _g1_point(a), "pi_b": _g2_point(b), "pi_c": _g1_point(c)}, pub, ms


def circuit_text(wc, key):
    """the "circuit" member of a final proof: what was proven, under which key"""
    if getattr(wc, "statement", None) is None:
        return ("final-stark-hashing+transcript (Merkle openings at the queries and the Fiat-Shamir sponge that places them; NOT the STARK's field "
                "arithmetic: the holder of the final STARK checks that): %d constraints (2^%d domain), %d wires, key %s (LOCAL SEEDED SETUP -- a test key, "
                "its trapdoor is public: not a proof under the reference's verifying key)" % (wc.c.n_constraints, wc.c.logm(), wc.c.n_wires, key.digest[:16]))
    return ("final-stark-verifier (stage B-2: Merkle openings, the Fiat-Shamir sponge, the constraint identity at zeta, the DEEP quotient and every fold of "
            "every query, the final layer's degree -- the public input commits to the statement's public inputs, zeta and the statement's sparse fixed "
            "columns at zeta, which a reader recomputes): %d constraints (2^%d domain), %d wires, key %s (LOCAL SEEDED SETUP -- a test key, its trapdoor "
            "is public: not a proof under the reference's verifying key)" % (wc.c.n_constraints, wc.c.logm(), wc.c.n_wires, key.digest[:16]))


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
