"""The part of a STARK verifier that needs NO OPENING, on the product side: parameters, the Fiat-Shamir transcript, the constraint identity at
the out-of-domain point, the low-degree test of the final FRI layer, the grinding check.

Why the prover service needs a verifier at all: GenFinalProof (proto/prover/v1/prover.proto:130-148; src/prover/provider.rs:472-503) takes the
aggregated proof as TEXT from the client.  The final STARK the service then makes proves the verifier AIR over that proof's queries -- every
authentication path, the DEEP quotient, every fold -- and building its witness fails when one of those fails; but the constraint identity at
zeta and the final layer's degree are not query-phase checks: without them a "proof" with an arbitrary trace and a low-degree quotient would be
wrapped into a pairing-valid Groth16 proof (round-4 advisor finding).  `verify_header` closes that: it is run on the aggregation STARK inside
Engine.final() before anything is proven.  The O(entries) part -- the ~4 x 10^5 sparse fixed-column entries of a verifier AIR at zeta -- is host
code of the library (csrc/verify.hip: zp_program_eval_ext); the transcript runs on the caller's backend (the same sponge the prover used).

The protocol mirrored here is stark/prover.py's (Goldilocks-hash mode); the independent statement of the same checks is the CPU checker's
oracle/stark_verify.py (header_only = True), which the tests run beside this one.  PARITY UNPINNED w.r.t. the external prover."""
from __future__ import annotations

from . import air as air_mod
from . import field as F
from .prover import PUBLICS_INLINE
from .transcript import Transcript

P = F.P


class Reject(ValueError):
    pass


def _mul_theta(a):                # theta (a0 + a1 t + a2 t^2), t^3 = t + 1: how three base columns carry one F_{p^3} column
    return [a[2] % P, (a[0] + a[2]) % P, a[1] % P]


def _e3_list(rows, n, what):
    if not isinstance(rows, list) or len(rows) != n:
        raise Reject("wrong number of %s" % what)
    out = []
    for r in rows:
        if not isinstance(r, list) or len(r) != 3 or any(not isinstance(v, int) or isinstance(v, bool) or not 0 <= v < P for v in r):
            raise Reject("malformed %s" % what)
        out.append([int(v) for v in r])
    return out


def verify_header(proof, air, params, be, root32=None, shift=None):
    """proof: a Goldilocks-mode proof OBJECT (its "queries" are not read); air: the statement (stark/air.py Air); params: the verifier's OWN
    StarkParams; be: a backend with poseidon_perm (and optionally poseidon_sponge, publics_digest).  Returns what the transcript dictates:
    {"indices", "zeta", "alpha", "gamma", "betas"}; raises Reject (a ValueError) with the reason otherwise."""
    from .. import native
    root32 = int(be.root32 if root32 is None else root32)
    shift = int(be.shift if shift is None else shift)
    if params.hash != "gl":
        raise Reject("verify_header reads Goldilocks-mode proofs")
    try:
        if proof["params"] != params.to_dict():
            raise Reject("proof was not made under the verifier's parameters")
        if proof["root32"] != root32 or proof["shift"] != shift:
            raise Reject("proof is over another evaluation domain")
        if proof["air_digest"] != air.digest():
            raise Reject("proof is for another statement")
        pubs = proof["publics"]
        if not isinstance(pubs, list) or len(pubs) != air.n_pub or any(not isinstance(v, int) or isinstance(v, bool) or not 0 <= v < P for v in pubs):
            raise Reject("malformed public inputs")
        logn, logb = params.logn, params.logb
        logm = logn + logb
        N, W, W2 = 1 << logn, air.width, air.width2
        Wt = W + W2
        Q = air_mod.quotient_chunks(air)
        if Q > (1 << logb):
            raise Reject("blow-up too small for the constraint degree")

        def root4(r):
            if not isinstance(r, list) or len(r) != 4 or any(not isinstance(v, int) or isinstance(v, bool) or not 0 <= v < P for v in r):
                raise Reject("malformed root")
            return [int(v) for v in r]

        tr = Transcript(be.poseidon_perm, getattr(be, "poseidon_sponge", None))
        head = [logn, logb, W, W2, params.fri_logf, params.fri_final_log, params.n_queries, params.pow_bits, root32, shift] + air.digest_words() + [len(pubs)]
        if len(pubs) <= PUBLICS_INLINE:
            tr.absorb(head + pubs)
        else:
            tr.absorb(head)
            dig = getattr(be, "publics_digest_gl", None) or be.publics_digest
            tr.absorb_root([int(v) for v in dig(pubs)])
        tr.absorb_root(root4(proof["roots"]["trace"]))
        chal = []
        if air.stage2:
            if "stage2" not in proof["roots"]:
                raise Reject("missing stage-2 commitment")
            chal = tr.challenge_e3()
            tr.absorb_root(root4(proof["roots"]["stage2"]))
        alpha = tr.challenge_e3()
        tr.absorb_root(root4(proof["roots"]["quotient"]))
        zeta = tr.challenge_e3()
        ev_all = _e3_list(proof["evals"]["z"], Wt + 3 * Q, "evaluations")
        ev_next = _e3_list(proof["evals"]["zw"], Wt, "evaluations")
        for r in ev_all + ev_next:
            tr.absorb(r)
        gamma = tr.challenge_e3()

        # ---- the constraint identity at zeta: sum_k alpha^k C_k(zeta) = q(zeta) Z_H(zeta)
        try:
            cs = native.program_eval_ext(air.program(), list(pubs) + list(chal), logn, root32, zeta, ev_all[:Wt], ev_next)
        except ValueError:
            raise Reject("the statement cannot be evaluated at the out-of-domain point")
        lhs, ap = [0, 0, 0], [1, 0, 0]
        for c in cs.tolist():
            lhs = F.e3_add(lhs, F.e3_mul(ap, [int(v) for v in c]))
            ap = F.e3_mul(ap, alpha)
        zh = F.e3_sub(F.e3_pow(zeta, N), [1, 0, 0])
        zsN = F.e3_pow(F.e3_scale(zeta, F.inv(shift)), N)                 # (zeta / shift)^N
        q, zpow = [0, 0, 0], [1, 0, 0]
        for j in range(Q):                                                 # q(zeta) = sum_j (zeta/shift)^(jN) (q_j0 + theta q_j1 + theta^2 q_j2)(zeta)
            qj = F.e3_add(F.e3_add(ev_all[Wt + 3 * j], _mul_theta(ev_all[Wt + 3 * j + 1])), _mul_theta(_mul_theta(ev_all[Wt + 3 * j + 2])))
            q = F.e3_add(q, F.e3_mul(zpow, qj))
            zpow = F.e3_mul(zpow, zsN)
        if [v % P for v in lhs] != [v % P for v in F.e3_mul(q, zh)]:
            raise Reject("constraint identity fails at the out-of-domain point")

        # ---- FRI transcript, final layer, grinding
        sched, final_log = params.fri_schedule()
        fr = proof["fri"]["roots"]
        if not isinstance(fr, list) or len(fr) != len(sched):
            raise Reject("wrong number of FRI layers")
        betas = []
        for r in fr:
            tr.absorb_root(root4(r))
            betas.append(tr.challenge_e3())
        final = proof["fri"]["final"]
        if not isinstance(final, list) or len(final) != 3 or any(not isinstance(pl, list) or len(pl) != (1 << final_log) or
                                                                  any(not isinstance(v, int) or isinstance(v, bool) or not 0 <= v < P for v in pl) for pl in final):
            raise Reject("bad final layer")
        for c in range(3):
            tr.absorb(final[c])
        if params.pow_bits:
            seed = tr.squeeze(4)
            nonce = proof.get("pow_nonce")
            if not isinstance(nonce, int) or isinstance(nonce, bool) or not 0 <= nonce < P:
                raise Reject("proof-of-work nonce missing or wrong")
            if be.poseidon_perm([int(v) for v in seed] + [nonce] + [0] * 7)[0] >> (64 - params.pow_bits):
                raise Reject("proof-of-work nonce missing or wrong")
            tr.absorb([nonce])
        indices = tr.indices(params.n_queries, logm)
        # the final layer is a polynomial of degree < 2^(final_log - logb) on its coset: interpolate (2^final_log values: tiny), unshift, look at the top
        s_final = shift
        for (_, f) in sched:
            s_final = pow(s_final, 1 << f, P)
        n_f = 1 << final_log
        w_inv, n_inv, s_inv = F.inv(F.root(final_log, root32)), F.inv(n_f), F.inv(s_final)
        keep = 1 << (final_log - logb)
        for c in range(3):
            for i in range(keep, n_f):                                     # coefficient i of the interpolant of plane c (times shift^-i: zero either way)
                wi, acc, cur = pow(w_inv, i, P), 0, 1
                for v in final[c]:
                    acc = (acc + v * cur) % P
                    cur = cur * wi % P
                if acc * n_inv % P * pow(s_inv, i, P) % P:
                    raise Reject("final FRI layer is not low degree")
        return {"indices": indices, "zeta": zeta, "alpha": alpha, "gamma": gamma, "betas": betas}
    except (KeyError, TypeError, IndexError) as e:
        raise Reject("malformed proof: %r" % (e,))
