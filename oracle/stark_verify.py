"""oracle/stark_verify.py -- independent verifier of the chunk STARK proofs (TEST INFRASTRUCTURE).

Written against the protocol description only (pure Python ints + oracle Poseidon), sharing no
code with the prover orchestration except the AIR's constraint expressions (the statement) and the
transcript rule.  A proof that verifies here is the end-to-end correctness check of the GPU path:
commitments open, the constraint identity holds out of domain, the DEEP quotient matches the opened
rows and every FRI fold is consistent down to a low-degree final layer."""
from __future__ import annotations

import numpy as np

from . import naive as NV
from . import oracle as O

P = NV.P


class Reject(Exception):
    pass


def _mul_theta(a):  # theta * (a0 + a1 t + a2 t^2), t^3 = t + 1
    return [a[2] % P, (a[0] + a[2]) % P, a[1] % P]


def _e3_sub(a, b):
    return [(a[i] - b[i]) % P for i in range(3)]


def _fold(values3, logf, beta, x_base, root32):
    """values3[k] (ext) = f(x_base * w_f^k); returns sum_j beta^j g_j(x_base^(2^logf))"""
    f = 1 << logf
    wf = NV.root(logf, root32) if logf else 1
    winv = pow(wf, P - 2, P)
    finv = pow(f, P - 2, P)
    xinv = pow(x_base, P - 2, P)
    acc, bp = [0, 0, 0], [1, 0, 0]
    for j in range(f):
        cj = [0, 0, 0]
        for k in range(f):
            w = pow(winv, (j * k) % f, P)
            cj = [(cj[c] + values3[k][c] * w) % P for c in range(3)]
        s = finv * pow(xinv, j, P) % P
        cj = [v * s % P for v in cj]
        acc = NV.e3_add(acc, NV.e3_mul(cj, bp))
        bp = NV.e3_mul(bp, beta)
    return acc


def verify(proof, air, rc, mds):
    from eigen_zeth_amd.stark.air import eval_constraints_ext
    from eigen_zeth_amd.stark.prover import StarkParams
    from eigen_zeth_amd.stark.transcript import Transcript

    rc = np.asarray(rc, dtype=np.uint64)
    mds = np.asarray(mds, dtype=np.uint64)
    params = StarkParams.from_dict(proof["params"])
    if proof["air"] != air.name or proof["air_digest"] != air.digest():
        raise Reject("proof is for a different AIR")
    root32, shift = proof["root32"], proof["shift"]
    logn, logb = params.logn, params.logb
    logm = logn + logb
    N, M, W = 1 << logn, 1 << logm, air.width
    pubs = proof["publics"]
    wN, wM = NV.root(logn, root32), NV.root(logm, root32)

    def perm(st):
        return [int(v) for v in O.poseidon_perm(np.array([st], dtype=np.uint64), rc, mds)[0]]

    tr = Transcript(perm)
    tr.absorb([logn, logb, W] + pubs)
    tr.absorb(proof["roots"]["trace"])
    W2 = air.width2
    Wt = W + W2
    chal = []
    if air.stage2:
        if "stage2" not in proof["roots"]:
            raise Reject("missing stage-2 commitment")
        chal = tr.challenge_e3()
        tr.absorb(proof["roots"]["stage2"])
    alpha = tr.challenge_e3()
    tr.absorb(proof["roots"]["quotient"])
    zeta = tr.challenge_e3()
    ev_all, ev_next = proof["evals"]["z"], proof["evals"]["zw"]
    if len(ev_all) != Wt + 3 or len(ev_next) != Wt:
        raise Reject("wrong number of evaluations")
    for r in ev_all + ev_next:
        tr.absorb(r)
    gamma = tr.challenge_e3()

    # ---- constraint identity at zeta
    zN = NV.e3_pow(zeta, N)
    zh = _e3_sub(zN, [1, 0, 0])
    ninv = pow(N, P - 2, P)
    wlast = pow(wN, N - 1, P)
    l_first = NV.e3_mul([v * ninv % P for v in zh], NV.e3_inv(_e3_sub(zeta, [1, 0, 0])))
    l_last = NV.e3_mul([v * ninv % P * wlast % P for v in zh], NV.e3_inv(_e3_sub(zeta, [wlast, 0, 0])))
    xml = _e3_sub(zeta, [wlast, 0, 0])
    cs = eval_constraints_ext(air, ev_all[:Wt], ev_next, [l_first, l_last], pubs, xml, NV.e3_mul, NV.e3_add, _e3_sub,
                              lambda v: [v % P, 0, 0], chal)
    lhs, ap = [0, 0, 0], [1, 0, 0]
    for c in cs:
        lhs = NV.e3_add(lhs, NV.e3_mul(ap, c))
        ap = NV.e3_mul(ap, alpha)
    q = ev_all[Wt]
    q = NV.e3_add(q, _mul_theta(ev_all[Wt + 1]))
    q = NV.e3_add(q, _mul_theta(_mul_theta(ev_all[Wt + 2])))
    if lhs != NV.e3_mul(q, zh):
        raise Reject("constraint identity fails at the out-of-domain point")

    # ---- FRI transcript
    sched, final_log = params.fri_schedule()
    if len(proof["fri"]["roots"]) != len(sched):
        raise Reject("wrong number of FRI layers")
    betas = []
    for root in proof["fri"]["roots"]:
        tr.absorb(root)
        betas.append(tr.challenge_e3())
    final = proof["fri"]["final"]
    if len(final) != 3 or any(len(pl) != (1 << final_log) for pl in final):
        raise Reject("bad final layer size")
    for c in range(3):
        tr.absorb(final[c])
    qidx = tr.indices(params.n_queries, logm)
    if [qq["index"] for qq in proof["queries"]] != qidx:
        raise Reject("query indices do not follow the transcript")

    # ---- final layer is low degree: degree < 2^(final_log - logb) on its coset
    s_final = shift
    for (_, f) in sched:
        s_final = pow(s_final, 1 << f, P)
    sinv = pow(s_final, P - 2, P)
    for c in range(3):
        cf = NV.intt(final[c], root32)
        cf = [cf[i] * pow(sinv, i, P) % P for i in range(len(cf))]
        if any(cf[(1 << (final_log - logb)):]):
            raise Reject("final FRI layer is not low degree")

    zeta_w = [v * wN % P for v in zeta]
    Wall = Wt + 3
    gp, cur = [], [1, 0, 0]
    for _ in range(Wall + Wt):
        gp.append(cur)
        cur = NV.e3_mul(cur, gamma)

    for qq in proof["queries"]:
        j = qq["index"]
        tv, qv = qq["trace"]["values"], qq["quotient"]["values"]
        if len(tv) != W or len(qv) != 3:
            raise Reject("bad opening width")
        if not O.merkle_verify(O.linear_hash(np.array(tv, dtype=np.uint64), rc, mds), M, j, np.array(qq["trace"]["path"], dtype=np.uint64),
                               np.array(proof["roots"]["trace"], dtype=np.uint64), rc, mds):
            raise Reject("trace opening does not verify")
        if not O.merkle_verify(O.linear_hash(np.array(qv, dtype=np.uint64), rc, mds), M, j, np.array(qq["quotient"]["path"], dtype=np.uint64),
                               np.array(proof["roots"]["quotient"], dtype=np.uint64), rc, mds):
            raise Reject("quotient opening does not verify")
        s2v = []
        if air.stage2:
            s2 = qq.get("stage2")
            if s2 is None or len(s2["values"]) != W2:
                raise Reject("missing stage-2 opening")
            s2v = s2["values"]
            if not O.merkle_verify(O.linear_hash(np.array(s2v, dtype=np.uint64), rc, mds), M, j, np.array(s2["path"], dtype=np.uint64),
                                   np.array(proof["roots"]["stage2"], dtype=np.uint64), rc, mds):
                raise Reject("stage-2 opening does not verify")
        x = shift * pow(wM, j, P) % P
        vals = tv + s2v + qv
        A, B = [0, 0, 0], [0, 0, 0]
        for k in range(Wall):
            A = NV.e3_add(A, NV.e3_mul(gp[k], _e3_sub([vals[k], 0, 0], ev_all[k])))
        for k in range(Wt):
            B = NV.e3_add(B, NV.e3_mul(gp[Wall + k], _e3_sub([vals[k], 0, 0], ev_next[k])))
        Fx = NV.e3_add(NV.e3_mul(A, NV.e3_inv(_e3_sub([x, 0, 0], zeta))),
                       NV.e3_mul(B, NV.e3_inv(_e3_sub([x, 0, 0], zeta_w))))
        # walk the layers
        expect, pos, cur_shift = Fx, j, shift
        if len(qq["fri"]) != len(sched):
            raise Reject("wrong number of FRI openings")
        for li, (lg, f) in enumerate(sched):
            m = 1 << (lg - f)
            row, k0 = pos & (m - 1), pos >> (lg - f)
            fo = qq["fri"][li]
            lv = fo["values"]
            if len(lv) != (3 << f):
                raise Reject("bad FRI leaf width")
            if not O.merkle_verify(O.linear_hash(np.array(lv, dtype=np.uint64), rc, mds), m, row, np.array(fo["path"], dtype=np.uint64),
                                   np.array(proof["fri"]["roots"][li], dtype=np.uint64), rc, mds):
                raise Reject("FRI opening does not verify (layer %d)" % li)
            pts = [[lv[c * (1 << f) + k] for c in range(3)] for k in range(1 << f)]
            if pts[k0] != expect:
                raise Reject("FRI layer %d value inconsistent with the previous layer" % li)
            x_base = cur_shift * pow(NV.root(lg, root32), row, P) % P
            expect = _fold(pts, f, betas[li], x_base, root32)
            pos = row
            cur_shift = pow(cur_shift, 1 << f, P)
        if [final[c][pos] for c in range(3)] != expect:
            raise Reject("final layer inconsistent with the last fold")
    return True
