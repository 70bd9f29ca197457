"""oracle/stark_verify.py -- independent verifier of the chunk STARK proofs (TEST INFRASTRUCTURE).

Written against the protocol description only (pure Python ints + oracle Poseidon).  It imports NOTHING from the
product package: the statement arrives as data (the constraint program blob, decoded by oracle/air_program.py), and
the Fiat-Shamir sponge, the parameter set and the FRI schedule are restated here.  A proof that verifies here is the
end-to-end correctness check of the GPU path: commitments open, the constraint identity holds out of domain, the DEEP
quotient matches the opened rows and every FRI fold is consistent down to a low-degree final layer.

The security parameters are the CALLER's (`expect`), never the proof's: a proof that claims other parameters (fewer
queries, another blow-up, another root of unity) is rejected, and every parameter is bound into the transcript.
PARITY UNPINNED with respect to the external reference prover (see gl_oracle.c)."""
from __future__ import annotations

import numpy as np

from . import naive as NV
from . import oracle as O
from .air_program import Program

P = NV.P
PARAM_KEYS = ("logn", "logb", "fri_logf", "fri_final_log", "n_queries", "pow_bits")


class Reject(Exception):
    pass


class Sponge:
    """Poseidon-12 sponge, rate 8 / capacity 4.  absorb() queues elements; the next squeeze first absorbs everything
    queued in blocks of 8 that overwrite the rate (zero padded; one permutation even when nothing is queued), then
    hands out the 8 rate elements in order, permuting again when they are used up."""

    def __init__(self, perm):
        self.perm, self.state, self.queue, self.avail = perm, [0] * 12, [], []

    def absorb(self, vals):
        self.queue.extend(int(v) % P for v in vals)
        self.avail = []

    def squeeze(self, n):
        out = []
        while len(out) < n:
            if self.queue or not self.avail:
                if not self.queue:
                    self.state = self.perm(self.state)
                while self.queue:
                    blk = self.queue[:8]
                    del self.queue[:8]
                    self.state = self.perm(blk + [0] * (8 - len(blk)) + self.state[8:])
                self.avail = list(self.state[:8])
            out.append(self.avail.pop(0))
        return out


class PublicSponge:
    """the sponge above with NO hashing: every permutation is vouched for by a verifier-AIR STARK whose public inputs hold, in
    protocol order, the values each permutation absorbs and the rate after every permutation the protocol reads
    (oracle/aggregate_verify.py).  `stream` is that list; absorbing compares, squeezing reads."""

    def __init__(self, stream):
        self.stream, self.pos, self.queue, self.avail = [int(v) for v in stream], 0, [], []

    def _take(self, n):
        if self.pos + n > len(self.stream):
            raise Reject("the public transcript is shorter than the protocol")
        out = self.stream[self.pos:self.pos + n]
        self.pos += n
        return out

    def absorb(self, vals):
        self.queue.extend(int(v) % P for v in vals)
        self.avail = []

    def squeeze(self, n):
        out = []
        while len(out) < n:
            if self.queue or not self.avail:
                while self.queue:
                    blk = self.queue[:8]
                    del self.queue[:8]
                    if self._take(len(blk)) != blk:
                        raise Reject("the public transcript absorbs other values than the proof's")
                self.avail = self._take(8)
            out.append(self.avail.pop(0))
        return out

    def pow_digest(self, seed4, nonce):
        """the grinding hash: seed || nonce in, the rate out"""
        if self._take(5) != [int(v) for v in seed4] + [int(nonce)]:
            raise Reject("the public grinding hash is over another seed or nonce")
        return self._take(8)


R_BN254 = NV.BN254_R


class SpongeBN128:
    """BN128-hash mode: Poseidon-BN254 sponge of width 17 (element 0 = capacity, 1..16 = rate).  absorb() packs Goldilocks
    values three to a field element (a + b 2^64 + c 2^128, each call padded separately), absorb_root() queues a root (one
    field element); squeeze() absorbs the queue in blocks of 16 overwriting the rate (one permutation even when empty) and
    hands out the three low 64-bit words of each rate element, reduced mod p."""

    def __init__(self, perm17):
        self.perm, self.state, self.queue, self.avail = perm17, [0] * 17, [], []

    def absorb(self, vals):
        v = [int(x) % P for x in vals]
        v += [0] * (-len(v) % 3)
        self.queue.extend(v[i] | (v[i + 1] << 64) | (v[i + 2] << 128) for i in range(0, len(v), 3))
        self.avail = []

    def absorb_root(self, root):
        if len(root) != 1 or not 0 <= int(root[0]) < R_BN254:
            raise Reject("malformed BN128 root")
        self.queue.append(int(root[0]))
        self.avail = []

    def squeeze(self, n):
        out = []
        while len(out) < n:
            if self.queue or not self.avail:
                if not self.queue:
                    self.state = self.perm(self.state)
                while self.queue:
                    blk = self.queue[:16]
                    del self.queue[:16]
                    self.state = self.perm([self.state[0]] + blk + [0] * (16 - len(blk)))
                self.avail = [((e >> s) & 0xFFFFFFFFFFFFFFFF) % P for e in self.state[1:] for s in (0, 64, 128)]
            out.append(self.avail.pop(0))
        return out


def merkle16_verify(leaf_digest, M, index, path, root):
    """path: per level the 16 digests of the group; the digest computed so far must sit at its position in the group"""
    n, pos, cur = M, int(index), int(leaf_digest)
    levels = 0
    while n > 1:
        if levels >= len(path) or len(path[levels]) != 16:
            return False
        grp = [int(v) for v in path[levels]]
        if any(not 0 <= v < R_BN254 for v in grp) or grp[pos % 16] != cur:
            return False
        g0 = (pos // 16) * 16
        if any(grp[c] != 0 for c in range(16) if g0 + c >= n):      # children beyond the level are zero
            return False
        cur = O.p254_perm([[0] + grp], 17)[0][0]
        n, pos, levels = (n + 15) // 16, pos // 16, levels + 1
    return levels == len(path) and cur == int(root)


PUBLICS_INLINE = 64


def publics_digest(pubs, rc, mds, bn):
    """more than PUBLICS_INLINE public inputs enter the transcript as one commitment.  Goldilocks mode: the values in rows of 8
    (zero padded, row count a power of two >= 2), leaf = linear hash of a row, binary Poseidon tree -> root (4 elements).  BN128
    mode: rows of 48 values (three to a field element, 16 elements to a leaf), 16-ary Poseidon-BN254 tree -> root (1 element)."""
    v = [int(x) % P for x in pubs]
    if not bn:
        M = 2
        while M * 8 < len(v):
            M <<= 1
        rows = np.zeros((M, 8), dtype=np.uint64)
        rows.reshape(-1)[:len(v)] = np.array(v, dtype=np.uint64)
        return [int(x) for x in O.merkle_commit_rows(rows, rc, mds)[-1]]
    M = max(1, -(-len(v) // 48))
    rows = np.zeros((M, 48), dtype=np.uint64)
    rows.reshape(-1)[:len(v)] = np.array(v, dtype=np.uint64)
    return [O._fr_ints(O.merkle16_tree(np.ascontiguousarray(rows.T))[-1])[0]]


def bn128_rows_per_leaf_log(width, logm):
    """BN128-hash mode: a leaf of a committed tree (trace, stage 2, quotient) holds 2^g rows i, i + M / 2^g, ...; g is the largest with
    width * 2^g <= 56 values (one width-17 permutation per leaf: naive.py pack_leaf_block), capped so that the tree keeps at least
    16 leaves"""
    g = 0
    while (width << (g + 1)) <= 56 and g + 1 <= logm - 4:
        g += 1
    return g


def _row_of_leaf(leaf, width, g, j, logm):
    """row j out of the 2^g rows a grouped leaf holds: column c of row j sits at position c 2^g + j / M'"""
    return [leaf[(c << g) + (j >> (logm - g))] for c in range(width)]


def fri_schedule(logn, logb, fri_logf, fri_final_log):
    """[(log size of the committed layer, log fold factor)], log size of the final layer sent in clear"""
    cur, stop, sched = logn + logb, fri_final_log + logb, []
    while cur > stop:
        f = min(fri_logf, cur - stop)
        sched.append((cur, f))
        cur -= f
    return sched, cur


def pow_ok(perm, seed4, nonce, bits):
    """grinding check: Poseidon(seed || nonce || 0^7)[0] has `bits` leading zero bits"""
    if bits == 0:
        return True
    out = perm([int(v) for v in seed4] + [int(nonce)] + [0] * 7)
    return 0 <= int(nonce) < P and (out[0] >> (64 - bits)) == 0


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


def expectation(params, root32=NV.ROOT32_DEFAULT, shift=NV.SHIFT_DEFAULT):
    """the verifier's parameter set from a plain dict of protocol parameters (+ the evaluation-domain constants)"""
    e = {k: int(params[k]) for k in PARAM_KEYS if k in params}
    e.setdefault("pow_bits", 0)
    if params.get("hash", "gl") != "gl":
        e["hash"] = params["hash"]
    e["root32"], e["shift"] = int(root32), int(shift)
    return e


def verify(proof, program, rc, mds, expect, bn_tables=None, header_only=False, trust_openings=False, public_transcript=None, trust_indices=False):
    """program: the constraint program blob (u64 words) of the statement;  expect: the verifier's own parameters
    {logn, logb, fri_logf, fri_final_log, n_queries, pow_bits, root32, shift [, hash]}.  hash = "bn128": the proof must be in
    BN128-hash mode (16-ary Poseidon-BN254 trees + transcript); bn_tables = (rc, mds, rp) of the t = 17 instance.
    header_only: check everything that needs no opening (parameters, transcript, out-of-domain constraint identity, final
    FRI layer, proof of work) and return {"indices": the query indices the transcript dictates} -- the part of the verifier
    that stays outside a Merkle-verifier AIR; proof["queries"] is not read.
    trust_openings: run the WHOLE verifier -- transcript, identity, DEEP quotient and every FRI fold at every query -- on the opened
    values as given, without their authentication paths (which a Merkle-verifier STARK vouches for: oracle/aggregate_verify.py);
    returns the same dictionary instead of True.
    trust_indices (with trust_openings): the query indices are taken as the proof states them, not compared with the transcript's -- for a
    proof whose Groth16 wrap hashes the transcript in its circuit and binds the indices there (oracle/wrap_verify.py: verify_rest).
    public_transcript (Goldilocks mode, with trust_openings): a PublicSponge -- the transcript is READ from public inputs a
    verifier-AIR STARK vouches for instead of being hashed here; the caller checks that the stream is used up."""
    rc = np.asarray(rc, dtype=np.uint64)
    mds = np.asarray(mds, dtype=np.uint64)
    air = program if isinstance(program, Program) else Program(program)
    missing = [k for k in PARAM_KEYS + ("root32", "shift") if k not in expect]
    if missing:
        raise ValueError("verifier parameters missing: %s" % missing)
    for k in PARAM_KEYS:
        if proof["params"].get(k) != expect[k]:
            raise Reject("proof claims %s = %r, the verifier requires %r" % (k, proof["params"].get(k), expect[k]))
    bn = expect.get("hash", "gl") == "bn128"
    if proof["params"].get("hash", "gl") != expect.get("hash", "gl"):
        raise Reject("proof is in another hash mode")
    if bn:
        if expect["pow_bits"] != 0:
            raise Reject("BN128 mode has no grinding")
        O.p254_set(17, bn_tables[2], bn_tables[0], bn_tables[1])
    if proof["root32"] != expect["root32"] or proof["shift"] != expect["shift"]:
        raise Reject("proof is over another evaluation domain")
    if expect["n_queries"] < 1 or expect["logb"] < 1:
        raise Reject("parameters give no soundness")
    if proof["air_digest"] != air.digest():
        raise Reject("proof is for a different AIR")
    root32, shift = expect["root32"], expect["shift"]
    logn, logb, n_queries, pow_bits = expect["logn"], expect["logb"], expect["n_queries"], expect["pow_bits"]
    if air.q_chunks > (1 << logb):
        raise Reject("blow-up too small for the constraint degree")
    logm = logn + logb
    N, M, W = 1 << logn, 1 << logm, air.width
    pubs = proof["publics"]
    if len(pubs) != air.n_pub:
        raise Reject("wrong number of public inputs")
    wN, wM = NV.root(logn, root32), NV.root(logm, root32)

    def perm(st):
        return [int(v) for v in O.poseidon_perm(np.array([st], dtype=np.uint64), rc, mds)[0]]

    W2 = air.width2
    Wt = W + W2
    if bn:
        tr = SpongeBN128(lambda st: O.p254_perm([st], 17)[0])
        absorb_root = lambda r: tr.absorb_root([int(v) for v in r])
        as_root = lambda r: int(r[0])

        def opening_ok(values, m, idx, path, root):
            return merkle16_verify(O.merkle16_leaf(np.array(values, dtype=np.uint64)), m, idx, path, as_root(root))
    else:
        tr = Sponge(perm) if public_transcript is None else public_transcript
        absorb_root = tr.absorb

        def opening_ok(values, m, idx, path, root):
            return bool(O.merkle_verify(O.linear_hash(np.array(values, dtype=np.uint64), rc, mds), m, idx, np.array(path, dtype=np.uint64),
                                        np.array(root, dtype=np.uint64), rc, mds))
    if trust_openings:
        opening_ok = lambda values, m, idx, path, root: all(0 <= int(v) < P for v in values)
    head = [logn, logb, W, W2, expect["fri_logf"], expect["fri_final_log"], n_queries, pow_bits, root32, shift] + air.digest_words() + [len(pubs)]
    if len(pubs) <= PUBLICS_INLINE:
        tr.absorb(head + pubs)
    else:
        tr.absorb(head)
        absorb_root(publics_digest(pubs, rc, mds, bn))
    absorb_root(proof["roots"]["trace"])
    chal = []
    if air.stage2:
        if "stage2" not in proof["roots"]:
            raise Reject("missing stage-2 commitment")
        chal = tr.squeeze(3)
        absorb_root(proof["roots"]["stage2"])
    alpha = tr.squeeze(3)
    absorb_root(proof["roots"]["quotient"])
    zeta = tr.squeeze(3)
    ev_all, ev_next = proof["evals"]["z"], proof["evals"]["zw"]
    Q = air.q_chunks                 # the quotient is committed as Q pieces of degree < N, 3 base columns each
    if len(ev_all) != Wt + 3 * Q or len(ev_next) != Wt:
        raise Reject("wrong number of evaluations")
    for r in ev_all + ev_next:
        tr.absorb(r)
    gamma = tr.squeeze(3)

    # ---- constraint identity at zeta
    zN = NV.e3_pow(zeta, N)
    zh = _e3_sub(zN, [1, 0, 0])
    ninv = pow(N, P - 2, P)
    wlast = pow(wN, N - 1, P)
    l_first = NV.e3_mul([v * ninv % P for v in zh], NV.e3_inv(_e3_sub(zeta, [1, 0, 0])))
    l_last = NV.e3_mul([v * ninv % P * wlast % P for v in zh], NV.e3_inv(_e3_sub(zeta, [wlast, 0, 0])))
    xml = _e3_sub(zeta, [wlast, 0, 0])
    fixed_z = [l_first, l_last] + [air.fixed_eval_ext(k, pubs, zeta, logn, root32) for k in range(len(air.fixed_cols))]
    cs = air.evaluate_ext(ev_all[:Wt], ev_next, fixed_z, list(pubs) + list(chal), xml)
    lhs, ap = [0, 0, 0], [1, 0, 0]
    for c in cs:
        lhs = NV.e3_add(lhs, NV.e3_mul(ap, c))
        ap = NV.e3_mul(ap, alpha)
    q, zpow = [0, 0, 0], [1, 0, 0]
    zsN = NV.e3_pow([v * pow(shift, P - 2, P) % P for v in zeta], N)     # (zeta / shift)^N
    for j in range(Q):               # q(zeta) = sum_j (zeta/shift)^(jN) (q_j0 + theta q_j1 + theta^2 q_j2)(zeta)
        qj = ev_all[Wt + 3 * j]
        qj = NV.e3_add(qj, _mul_theta(ev_all[Wt + 3 * j + 1]))
        qj = NV.e3_add(qj, _mul_theta(_mul_theta(ev_all[Wt + 3 * j + 2])))
        q = NV.e3_add(q, NV.e3_mul(zpow, qj))
        zpow = NV.e3_mul(zpow, zsN)
    if lhs != NV.e3_mul(q, zh):
        raise Reject("constraint identity fails at the out-of-domain point")

    # ---- FRI transcript
    sched, final_log = fri_schedule(logn, logb, expect["fri_logf"], expect["fri_final_log"])
    if len(proof["fri"]["roots"]) != len(sched):
        raise Reject("wrong number of FRI layers")
    betas = []
    for root in proof["fri"]["roots"]:
        absorb_root(root)
        betas.append(tr.squeeze(3))
    final = proof["fri"]["final"]
    if len(final) != 3 or any(len(pl) != (1 << final_log) for pl in final):
        raise Reject("bad final layer size")
    for c in range(3):
        tr.absorb(final[c])
    if pow_bits:
        seed = tr.squeeze(4)
        nonce = proof.get("pow_nonce")
        if nonce is None or not 0 <= int(nonce) < P:
            raise Reject("proof-of-work nonce missing or wrong")
        if (public_transcript is None or bn) and not pow_ok(perm, seed, nonce, pow_bits):
            raise Reject("proof-of-work nonce missing or wrong")
        tr.absorb([nonce])
    qidx = [v & (M - 1) for v in tr.squeeze(n_queries)]
    if pow_bits and public_transcript is not None and not bn:
        # the grinding hash is the last permutation of a proof's public transcript: seed || nonce in, digest out
        if public_transcript.pow_digest(seed, nonce)[0] >> (64 - pow_bits):
            raise Reject("proof-of-work nonce missing or wrong")
    if trust_indices and not trust_openings:
        raise ValueError("trust_indices is for openings a circuit vouches for")
    if trust_indices:
        if len(proof["queries"]) != n_queries or any(not isinstance(qq["index"], int) or not 0 <= qq["index"] < M for qq in proof["queries"]):
            raise Reject("malformed query indices")
        qidx = [qq["index"] for qq in proof["queries"]]
    elif not header_only and (len(proof["queries"]) != n_queries or [qq["index"] for qq in proof["queries"]] != qidx):
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

    if header_only:
        return {"indices": qidx, "sched": sched, "W": W, "W2": W2, "Wq": 3 * Q, "zeta": zeta, "gamma": gamma, "betas": betas}
    zeta_w = [v * wN % P for v in zeta]
    Wall = Wt + 3 * Q
    gp, cur = [], [1, 0, 0]
    for _ in range(Wall + Wt):
        gp.append(cur)
        cur = NV.e3_mul(cur, gamma)

    # BN128 mode: 2^g rows per leaf of the trace / stage-2 / quotient trees
    gt, g2, qg = [(bn128_rows_per_leaf_log(w, logm) if bn and w else 0) for w in (W, W2, 3 * Q)]
    for qq in proof["queries"]:
        j = qq["index"]
        tleaf, qleaf = qq["trace"]["values"], qq["quotient"]["values"]
        if len(tleaf) != W << gt or len(qleaf) != (3 * Q) << qg:
            raise Reject("bad opening width")
        if not opening_ok(tleaf, M >> gt, j & ((M >> gt) - 1), qq["trace"].get("path"), proof["roots"]["trace"]):
            raise Reject("trace opening does not verify")
        if not opening_ok(qleaf, M >> qg, j & ((M >> qg) - 1), qq["quotient"].get("path"), proof["roots"]["quotient"]):
            raise Reject("quotient opening does not verify")
        tv, qv = _row_of_leaf(tleaf, W, gt, j, logm), _row_of_leaf(qleaf, 3 * Q, qg, j, logm)
        s2v = []
        if air.stage2:
            s2 = qq.get("stage2")
            if s2 is None or len(s2["values"]) != W2 << g2:
                raise Reject("missing stage-2 opening")
            if not opening_ok(s2["values"], M >> g2, j & ((M >> g2) - 1), s2.get("path"), proof["roots"]["stage2"]):
                raise Reject("stage-2 opening does not verify")
            s2v = _row_of_leaf(s2["values"], W2, g2, j, logm)
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
            if not opening_ok(lv, m, row, fo.get("path"), proof["fri"]["roots"][li]):
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
    return {"indices": qidx, "sched": sched, "W": W, "W2": W2, "Wq": 3 * Q} if trust_openings else True
