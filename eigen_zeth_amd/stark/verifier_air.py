"""The STARK-verifier AIR, stage A: the Merkle part of verifying chunk proofs, as constraints.

Serves GenAggregatedProof (proto/prover/v1/prover.proto:115-126; client src/prover/provider.rs:422-451) and the final STARK
of GenFinalProof (prover.proto:130-148; provider.rs:472-503): "aggregate(p1, p2)" is a STARK whose witness is the
verification trace of the inner proofs' query openings -- every Poseidon permutation a verifier runs to check that the
opened rows of the trace / stage-2 / quotient / FRI-layer commitments hash up to the roots the proofs name -- and whose
public inputs are those roots and the query indices.  The reference holds no prover arithmetic, so the
construction is this repo's own (parity unpinned); it follows the public recursive-STARK recipe.

What the AIR proves (public inputs: per inner proof and tree the root; per query slot / proof / tree the leaf index and the
OPENED VALUES):
    for every query slot, inner proof and committed tree there is an authentication path such that linear_hash(the PUBLIC
    leaf values) hashed up the path, with the direction bits of the PUBLIC index, equals the PUBLIC root.
That is ALL the hashing of a verifier.  What is left of verifying an inner proof is arithmetic on public data -- replaying the
Fiat-Shamir transcript (which yields the indices), the out-of-domain constraint identity, the DEEP quotient and every FRI fold
at every query on the opened values, the final layer -- and the checker of an aggregated proof does exactly that natively
(oracle/aggregate_verify.py: stark_verify.verify(trust_openings=True) on the inner proofs WITHOUT their paths) and requires
the outer proof's public inputs to be those roots, indices and values.  So an accepted aggregated proof means both inner proofs
verify: the Merkle work in the circuit, the field arithmetic outside it (putting that arithmetic into constraints too is what
would make the recursion succinct in it: DESIGN.md par.7).

The Fiat-Shamir transcripts of the inner proofs are in the circuit too (stage B, hashing part): every permutation of an inner
proof's sponge is a block of the trace, chained through the capacity (or the whole state, between squeezes), with the absorbed
blocks and the squeezed rates as public inputs -- and the grinding hash, seed and nonce in, digest out.  The checker replays a
transcript WITHOUT hashing: it compares the blocks the protocol absorbs (parameters, AIR digest, public inputs or their digest,
roots, out-of-domain evaluations, final layer, nonce) with the public ones and reads the challenges, the grinding digest and the
query indices off the public rates (oracle/aggregate_verify.py).  A verifier of an aggregated proof hashes nothing but the
public-input digest of the outer proof itself.

Layout.  One Poseidon-12 permutation = one BLOCK of 32 rows: row r < 30 holds the state before round r, row 30 the
output, row 31 a copy of it; the cubes (s_i + rc_i)^3 sit in 12 helper columns so that x^7 = cube^2 * x has degree 3.  Full
and partial rounds share one constraint through the periodic selector FULL (degree 4 -> quotient in 3 pieces, blow-up 4).
The link between block k (row 31) and block k+1 (row 0) is chosen by SCHEDULE selectors (sparse periodic fixed columns,
stark/air.py FixedCol): sponge chaining of a leaf hash, a Merkle node (left / right by the direction bit, accumulated into the
index), the final comparison with the root and the index.  The schedule is fixed by the shape of the inner proofs, so a
prover cannot shorten a path or skip a comparison.  Everything is cyclic (no boundary constraints): the last link of the
trace wraps to block 0."""
from __future__ import annotations

import numpy as np

from . import air as A
from .air import Col, Fixed, FixedCol, Pub, Const

P = A.P
ROWS = 32          # rows per permutation block
N_ROUNDS, N_FULL_HALF = 30, 4
S0, U0, COL_D, COL_IDX, WIDTH = 0, 12, 24, 25, 26


class Shape:
    """what the Merkle part of a verifier needs to know about the inner proofs (all inner proofs of one aggregation share it)"""

    def __init__(self, logn, logb, W, W2, Wq, n_queries, fri_logf, fri_final_log, n_proofs=2, n_pub_inner=0, pow_bits=0):
        """n_pub_inner / pow_bits: number of public inputs and grinding bits of an inner proof (they shape its transcript)"""
        self.logn, self.logb, self.W, self.W2, self.Wq = logn, logb, W, W2, Wq
        self.n_queries, self.fri_logf, self.fri_final_log, self.n_proofs = n_queries, fri_logf, fri_final_log, n_proofs
        self.n_pub_inner, self.pow_bits = n_pub_inner, pow_bits
        logm = logn + logb
        self.final_log = logm
        self.trees = [("trace", W, logm)]
        if W2:
            self.trees.append(("stage2", W2, logm))
        self.trees.append(("quotient", Wq, logm))
        cur, stop, li = logm, fri_final_log + logb, 0
        while cur > stop:
            f = min(fri_logf, cur - stop)
            self.trees.append(("fri%d" % li, 3 << f, cur - f))
            cur -= f
            li += 1
        self.final_log = cur                       # log2 of the FRI layer sent in clear
        self.n_fri = li
        assert all(d >= 1 for (_, _, d) in self.trees)

    @staticmethod
    def of_proof(proof, n_proofs=2):
        """shape of a proof object (dict as stark/prover.py writes it)"""
        pr = proof["params"]
        q0 = proof["queries"][0]
        return Shape(pr["logn"], pr["logb"], len(q0["trace"]["values"]), len(q0["stage2"]["values"]) if "stage2" in q0 else 0,
                     len(q0["quotient"]["values"]), pr["n_queries"], pr["fri_logf"], pr["fri_final_log"], n_proofs,
                     len(proof["publics"]), pr["pow_bits"])

    KEY_NAMES = ("logn", "logb", "W", "W2", "Wq", "n_queries", "fri_logf", "fri_final_log", "n_proofs", "n_pub_inner", "pow_bits")

    def key(self):
        return (self.logn, self.logb, self.W, self.W2, self.Wq, self.n_queries, self.fri_logf, self.fri_final_log, self.n_proofs,
                self.n_pub_inner, self.pow_bits)

    def to_dict(self):
        return dict(zip(self.KEY_NAMES, self.key()))

    @staticmethod
    def from_dict(d):
        """shape named by an aggregated proof (untrusted text: bounded before anything is sized by it)"""
        v = [d[k] for k in Shape.KEY_NAMES]
        lim = dict(zip(Shape.KEY_NAMES, (30, 8, 4096, 4096, 64, 4096, 8, 16, 64, 1 << 24, 64)))
        if not all(isinstance(x, int) and not isinstance(x, bool) and 0 <= x <= lim[k] for k, x in zip(Shape.KEY_NAMES, v)):
            raise ValueError("shape out of range")
        if v[0] < 1 or v[1] < 1 or v[2] < 1 or v[4] < 1 or v[5] < 1 or v[6] < 1 or v[8] < 1 or v[0] + v[1] > 32:
            raise ValueError("shape out of range")
        return Shape(*v)

    # ---- the Fiat-Shamir transcript of one inner proof, as a list of permutations
    def transcript_perms(self):
        """the permutations of ONE inner proof's transcript in the order the protocol runs them (stark/prover.py; the sponge of
        stark/transcript.py: absorb queues, a squeeze first absorbs what is queued in blocks of 8 that overwrite the rate -- one
        bare permutation if nothing is queued -- and permutes again when its 8 outputs are used up).  List of dicts: n_in =
        values absorbed by this permutation's block (0: the whole state is carried), first (the state before it is zero), out
        (its rate is read by the protocol), pin / pout = offsets of its absorbed values / its 8 rate outputs in the proof's
        section of the public inputs.  With grinding the LAST entry is the grinding hash (pow: seed || nonce in, digest out)."""
        cached = _SCRIPT_CACHE.get(self.key())
        if cached is not None:
            return cached
        from .prover import PUBLICS_INLINE
        perms, st = [], {"queue": 0, "avail": 0}

        def absorb(n):
            st["queue"] += n
            st["avail"] = 0

        def squeeze(n):
            for _ in range(n):
                if st["queue"] or not st["avail"]:
                    if not st["queue"]:
                        perms.append({"n_in": 0})
                    while st["queue"]:
                        b = min(8, st["queue"])
                        st["queue"] -= b
                        perms.append({"n_in": b})
                    perms[-1]["out"] = True
                    st["avail"] = 8
                st["avail"] -= 1
        Wt = self.W + self.W2
        absorb(10 + 4 + 1 + (self.n_pub_inner if self.n_pub_inner <= PUBLICS_INLINE else 4))
        absorb(4)
        if self.W2:
            squeeze(3)
            absorb(4)
        squeeze(3)
        absorb(4)
        squeeze(3)
        absorb(3 * (Wt + self.Wq) + 3 * Wt)
        squeeze(3)
        for _ in range(self.n_fri):
            absorb(4)
            squeeze(3)
        absorb(3 << self.final_log)
        if self.pow_bits:
            squeeze(4)
            absorb(1)
        squeeze(self.n_queries)
        if self.pow_bits:
            perms.append({"n_in": 5, "out": True, "pow": True})
        off = 0
        for j, pm in enumerate(perms):
            pm.setdefault("out", False)
            pm.setdefault("pow", False)
            pm["first"] = j == 0 or pm["pow"]
            pm["pin"] = off
            off += pm["n_in"]
            if pm["out"]:
                pm["pout"] = off
                off += 8
        _SCRIPT_CACHE[self.key()] = perms
        return perms

    def transcript_pubs_per_proof(self):
        last = self.transcript_perms()[-1]
        return last["pout"] + 8            # the last permutation of a transcript (queries or grinding) is always read

    def transcript_block0(self):
        """global block number of the first transcript block: the idle tail of the LAST period"""
        k, periods, pb = self.layout()
        return (periods - 1) * pb + k * self.n_proofs * self.blocks_per_proof()

    def merkle_pubs(self):
        T = len(self.trees)
        return self.n_proofs * T * 4 + self.n_slots() * self.n_proofs * (T + self.values_per_query())

    def pub_tin(self, p, j, i):
        return self.merkle_pubs() + p * self.transcript_pubs_per_proof() + self.transcript_perms()[j]["pin"] + i

    def pub_tout(self, p, j, i):
        return self.merkle_pubs() + p * self.transcript_pubs_per_proof() + self.transcript_perms()[j]["pout"] + i

    # ---- the block schedule
    @staticmethod
    def absorb_blocks(width):
        return 0 if width <= 4 else -(-width // 8)

    def blocks_per_proof(self):
        return sum(self.absorb_blocks(w) + d for (_, w, d) in self.trees)

    def layout(self):
        """(k, periods, PB): k query slots per period, `periods` periods (a power of two), PB blocks per period (a power of
        two >= k * n_proofs * blocks_per_proof); chosen to minimise the trace length 32 * PB * periods"""
        bg = self.n_proofs * self.blocks_per_proof()
        tb = self.n_proofs * len(self.transcript_perms())      # transcript blocks: in the idle tail of the last period
        best = None
        for j in range(0, 12):
            periods = 1 << j
            k = -(-self.n_queries // periods)
            pb = 1
            while pb < k * bg + tb:
                pb <<= 1
            rows = ROWS * pb * periods
            if best is None or rows < best[0]:
                best = (rows, k, periods, pb)
            if k == 1:
                break
        return best[1], best[2], best[3]

    def logn_trace(self):
        k, periods, pb = self.layout()
        return (ROWS * pb * periods).bit_length() - 1

    def n_slots(self):
        k, periods, _ = self.layout()
        return k * periods

    def n_pub(self):
        return self.merkle_pubs() + self.n_proofs * self.transcript_pubs_per_proof()

    def pub_root(self, p, t, i):
        return (p * len(self.trees) + t) * 4 + i

    def pub_index(self, slot, p, t):
        T = len(self.trees)
        return self.n_proofs * T * 4 + (slot * self.n_proofs + p) * T + t

    def values_per_query(self):
        return sum(w for (_, w, _) in self.trees)

    def pub_value(self, slot, p, t, k):
        """public input number of opened value k of tree t of proof p in query slot `slot`"""
        T = len(self.trees)
        base = self.n_proofs * T * 4 + self.n_slots() * self.n_proofs * T
        return base + (slot * self.n_proofs + p) * self.values_per_query() + sum(w for (_, w, _) in self.trees[:t]) + k

    def period_schedule(self):
        """blocks of one period: list of dicts {kind: "absorb" | "node" | "idle", sub (slot within the period), p, t, level, first,
        last}; idle blocks pad the period to PB"""
        cached = _SCHED_CACHE.get(self.key())
        if cached is not None:
            return cached
        k, _, pb = self.layout()
        out = []
        for sub in range(k):
            for p in range(self.n_proofs):
                for t, (_, w, d) in enumerate(self.trees):
                    a = self.absorb_blocks(w)
                    for j in range(a):
                        out.append({"kind": "absorb", "sub": sub, "p": p, "t": t, "j": j, "first": j == 0, "last": False})
                    for lv in range(d):
                        out.append({"kind": "node", "sub": sub, "p": p, "t": t, "level": lv, "first": a == 0 and lv == 0, "last": lv == d - 1})
        out += [{"kind": "idle", "first": False, "last": False}] * (pb - len(out))
        _SCHED_CACHE[self.key()] = out          # tens of thousands of entries, fixed by the shape: built once (callers only read it)
        return out


_AIR_CACHE = {}
_SCHED_CACHE = {}
_SCRIPT_CACHE = {}


def verifier_air(shape, rc, mds):
    """the AIR for inner proofs of `shape` under the Poseidon tables rc (360) / mds (144, row-major)"""
    key = (shape.key(), hash(tuple(int(v) for v in rc)), hash(tuple(int(v) for v in mds)))
    if key in _AIR_CACHE:
        return _AIR_CACHE[key]
    rc = [int(v) % P for v in rc]
    mds = [int(v) % P for v in mds]
    k, periods, pb = shape.layout()
    lp = (ROWS * pb).bit_length() - 1            # log2 of the schedule period in rows
    logn = shape.logn_trace()
    sched = shape.period_schedule()
    T = len(shape.trees)

    # ---- fixed columns
    fc = []
    for i in range(12):
        fc.append(FixedCol(5, [(r, rc[r * 12 + i]) for r in range(N_ROUNDS) if rc[r * 12 + i]]))
    full_rows = [r for r in range(N_ROUNDS) if r < N_FULL_HALF or r >= N_ROUNDS - N_FULL_HALF]
    fc.append(FixedCol(5, [(r, 1) for r in full_rows]))                   # FULL
    fc.append(FixedCol(5, [(r, 1) for r in range(N_ROUNDS)]))             # ACT: rows with a round transition
    fc.append(FixedCol(5, [(30, 1)]))                                     # CPY: row 30 -> 31 copies the output
    RC = [Fixed(2 + i) for i in range(12)]
    FULL, ACT, CPY = Fixed(14), Fixed(15), Fixed(16)
    l_chain, l_node, l_final, cap0, wt = [], [], [], [], []
    roots = [[] for _ in range(4)]
    final_rows = []                                                       # (row in period, sub, p, t)
    for b, blk in enumerate(sched):
        row = b * ROWS + 31
        nxt = sched[(b + 1) % len(sched)]
        if blk["kind"] == "absorb":
            (l_chain if nxt["kind"] == "absorb" and not nxt["first"] else l_node).append((row, 1))
        elif blk["kind"] == "node":
            if blk["last"]:
                l_final.append((row, 1))
                for i in range(4):
                    roots[i].append((row, Pub(shape.pub_root(blk["p"], blk["t"], i))))
                final_rows.append((row, blk["sub"], blk["p"], blk["t"]))
            else:
                l_node.append((row, 1))
        if nxt["kind"] == "node" or nxt["first"]:
            cap0.append((row, 1))                                         # capacity of the next block's input is zero
        if nxt["kind"] == "node":
            wt.append((row, 1 << nxt["level"]))                           # weight of the next block's direction bit
    fc += [FixedCol(lp, l_chain), FixedCol(lp, l_node), FixedCol(lp, l_final), FixedCol(lp, cap0), FixedCol(lp, wt)]
    L_CHAIN, L_NODE, L_FINAL, CAP0, WT = [Fixed(17 + i) for i in range(5)]
    fc += [FixedCol(lp, roots[i]) for i in range(4)]
    ROOT = [Fixed(22 + i) for i in range(4)]
    idx_entries = []
    for per in range(periods):
        for (row, sub, p, t) in final_rows:
            idx_entries.append((per * (ROWS * pb) + row, Pub(shape.pub_index(per * k + sub, p, t))))
    fc.append(FixedCol(logn, idx_entries))
    IDXV = Fixed(26)
    # the opened values are public: row 0 of absorb block j holds values 8j .. 8j+7 of the leaf in the rate (positions past the
    # leaf width are zero: the padding of the sponge), row 0 of the first node block of an unhashed leaf (width <= 4) holds it as
    # the left or right child
    abs0, id0 = [], []
    leaf = [[] for _ in range(8)]
    for per in range(periods):
        for b, blk in enumerate(sched):
            if blk["kind"] == "idle":
                continue
            row = per * (ROWS * pb) + b * ROWS
            w = shape.trees[blk["t"]][1]
            slot = per * k + blk["sub"]
            if blk["kind"] == "absorb":
                for i in range(8):
                    if 8 * blk["j"] + i < w:
                        leaf[i].append((row, Pub(shape.pub_value(slot, blk["p"], blk["t"], 8 * blk["j"] + i))))
            elif blk["first"]:
                for i in range(w):
                    leaf[i].append((row, Pub(shape.pub_value(slot, blk["p"], blk["t"], i))))
    for b, blk in enumerate(sched):
        if blk["kind"] == "absorb":
            abs0.append((b * ROWS, 1))
        elif blk["kind"] == "node" and blk["first"]:
            id0.append((b * ROWS, 1))
    # the transcript blocks (idle tail of the last period, one run of consecutive blocks per inner proof): links through the
    # capacity (absorbing permutation) or the whole state (a permutation between two squeezes), zero capacity before the first
    # permutation and before the grinding hash; the absorbed blocks (row 0) and the rates the protocol reads (row 31) are public
    tcap, trate, tcap0, tabs = [], [], [], []
    perms = shape.transcript_perms()
    for p in range(shape.n_proofs):
        for j, pm in enumerate(perms):
            row0 = ROWS * (shape.transcript_block0() + p * len(perms) + j)
            if pm["first"]:
                tcap0.append((row0 - 1, 1))
            else:
                tcap.append((row0 - 1, 1))
                if pm["n_in"] == 0:
                    trate.append((row0 - 1, 1))
            if pm["n_in"]:
                tabs.append((row0, 1))
                for i in range(pm["n_in"]):
                    leaf[i].append((row0, Pub(shape.pub_tin(p, j, i))))
            if pm["out"]:
                tabs.append((row0 + 31, 1))
                for i in range(8):
                    leaf[i].append((row0 + 31, Pub(shape.pub_tout(p, j, i))))
    fc += [FixedCol(lp, abs0), FixedCol(lp, id0)] + [FixedCol(logn, leaf[i]) for i in range(8)]
    ABS0, ID0 = Fixed(27), Fixed(28)
    LEAF = [Fixed(29 + i) for i in range(8)]
    fc += [FixedCol(logn, tcap), FixedCol(logn, trate), FixedCol(logn, tcap0), FixedCol(logn, tabs)]
    TCAP, TRATE, TCAP0, TABS = [Fixed(37 + i) for i in range(4)]

    # ---- constraints
    s = [Col(S0 + i) for i in range(12)]
    sn = [Col(S0 + i, True) for i in range(12)]
    u = [Col(U0 + i) for i in range(12)]
    d_c, d_n, idx, idx_n = Col(COL_D), Col(COL_D, True), Col(COL_IDX), Col(COL_IDX, True)
    x = [s[i] + RC[i] for i in range(12)]
    cs = [u[i] - x[i] * x[i] * x[i] for i in range(12)]                   # cubes (every row; rc = 0 on rows 30, 31)
    y = [ACT * (u[0] * u[0] * x[0])]
    for i in range(1, 12):                                                # S-box output: x^7 in full rounds, x in partial rounds (element 0: always x^7)
        y.append(ACT * x[i] + FULL * (x[i] * (u[i] * u[i]) - x[i]))
    for j in range(12):
        acc = None
        for i in range(12):
            m = mds[j * 12 + i]
            if m == 0:
                continue
            term = y[i] if m == 1 else Const(m) * y[i]
            acc = term if acc is None else acc + term
        cs.append(ACT * sn[j] - acc)                                      # round transition on rows 0..29
    cs += [CPY * (sn[j] - s[j]) for j in range(12)]                       # row 31 = row 30
    cs += [L_CHAIN * (sn[8 + i] - s[i]) for i in range(4)]                # sponge chaining: the digest becomes the next capacity
    cs += [L_NODE * (sn[i] - s[i] + d_n * (sn[4 + i] - sn[i])) for i in range(4)]   # digest = left (bit 0) or right (bit 1) child
    cs += [(CAP0 + TCAP0) * sn[8 + i] for i in range(4)]
    cs += [TCAP * (sn[8 + i] - s[8 + i]) for i in range(4)]               # transcript sponge: the capacity runs through the permutations
    cs += [TRATE * (sn[i] - s[i]) for i in range(8)]                      # ... and the rate too when nothing is absorbed in between
    cs.append(WT * (d_n * d_n - d_n))                                     # direction bits are bits
    cs.append(idx_n - (ACT + CPY + L_CHAIN + L_NODE) * idx - WT * d_n)    # index: kept inside a block and an opening, + 2^level * bit
    cs += [L_FINAL * s[i] - ROOT[i] for i in range(4)]                    # the top of the path is the public root
    cs.append(L_FINAL * idx - IDXV)                                       # the direction bits spell the public index
    # the hashed values are the public ones (leaves; absorbed blocks and read rates of the transcripts)
    cs += [(ABS0 + TABS) * s[i] + ID0 * (s[i] + d_c * (s[4 + i] - s[i])) - LEAF[i] for i in range(4)]
    cs += [(ABS0 + TABS) * s[i] - LEAF[i] for i in range(4, 8)]
    air = A.Air("mverify", WIDTH, shape.n_pub(), cs, trace_kind=None, fixed_cols=fc)
    air.shape = shape
    assert A.quotient_chunks(air) <= 4
    _AIR_CACHE[key] = air
    return air


# ---------------------------------------------------------------------------------------------------------------- witness
def _opening(q, name):
    if name.startswith("fri"):
        return q["fri"][int(name[3:])]
    return q[name]


class _RecordingSponge:
    """the sponge of stark/transcript.py that keeps, for every permutation, the state it started from: [input state (12), values
    absorbed by its block (0: state carried), the rate after it if the protocol reads it (else None)].  One device call per
    transcript step when the backend has poseidon_sponge_caps (zp_poseidon_sponge_caps), else permutation by permutation."""

    def __init__(self, be):
        self.be, self.state, self.queue, self.avail, self.rec = be, [0] * 12, [], [], []

    def absorb(self, vals):
        self.queue += [int(v) % P for v in vals]
        self.avail = []

    def _flush(self):
        blocks = [self.queue[i:i + 8] for i in range(0, len(self.queue), 8)]
        self.queue = []
        nin = [len(b) for b in blocks] or [0]
        blocks = [b + [0] * (8 - len(b)) for b in blocks]
        if hasattr(self.be, "poseidon_sponge_caps"):
            new_state, _, caps = self.be.poseidon_sponge_caps(self.state, blocks, 0)
            if blocks:
                inputs = [blocks[i] + [int(v) for v in (self.state[8:] if i == 0 else caps[i - 1])] for i in range(len(blocks))]
            else:
                inputs = [list(self.state)]
            self.state = [int(v) for v in new_state]
        else:
            inputs = []
            if not blocks:
                inputs.append(list(self.state))
                self.state = [int(v) for v in self.be.poseidon_perm(self.state)]
            for b in blocks:
                inputs.append(b + list(self.state[8:]))
                self.state = [int(v) for v in self.be.poseidon_perm(inputs[-1])]
        for inp, n in zip(inputs, nin):
            self.rec.append([inp, n, None])
        self.rec[-1][2] = list(self.state[:8])
        self.avail = list(self.state[:8])

    def squeeze(self, n):
        out = []
        while len(out) < n:
            if self.queue or not self.avail:
                self._flush()
            out.append(self.avail.pop(0))
        return out


def replay_transcript(shape, proof, digest_words, be):
    """the Fiat-Shamir transcript of an inner proof, permutation by permutation: (input states [L][12] in the order of
    shape.transcript_perms(), the proof's section of the public inputs).  digest_words: the inner AIR's digest as the prover
    absorbs it (Air.digest_words()).  Raises ValueError when the transcript does not produce the proof's query indices or the
    grinding nonce fails -- such a proof has no accepting witness."""
    from .prover import PUBLICS_INLINE
    pr = proof["params"]
    pubs = [int(v) for v in proof["publics"]]
    try:
        tr = _RecordingSponge(be)
        head = [pr["logn"], pr["logb"], shape.W, shape.W2, pr["fri_logf"], pr["fri_final_log"], pr["n_queries"], pr["pow_bits"],
                int(proof["root32"]), int(proof["shift"])] + [int(v) for v in digest_words] + [len(pubs)]
        if len(pubs) <= PUBLICS_INLINE:
            tr.absorb(head + pubs)
        else:
            tr.absorb(head)
            tr.absorb(be.publics_digest_gl(pubs))
        tr.absorb(proof["roots"]["trace"])
        if shape.W2:
            tr.squeeze(3)
            tr.absorb(proof["roots"]["stage2"])
        tr.squeeze(3)
        tr.absorb(proof["roots"]["quotient"])
        tr.squeeze(3)
        for r in proof["evals"]["z"] + proof["evals"]["zw"]:
            tr.absorb(r)
        tr.squeeze(3)
        for root in proof["fri"]["roots"]:
            tr.absorb(root)
            tr.squeeze(3)
        for c in range(3):
            tr.absorb(proof["fri"]["final"][c])
        pow_rec = None
        if shape.pow_bits:
            seed = tr.squeeze(4)
            nonce = int(proof.get("pow_nonce", -1))
            if not 0 <= nonce < P:
                raise ValueError("grinding nonce missing")
            pin = seed + [nonce] + [0] * 7
            pout = [int(v) for v in be.poseidon_perm(pin)]
            if pout[0] >> (64 - shape.pow_bits):
                raise ValueError("the grinding nonce of an inner proof is wrong: no accepting witness")
            pow_rec = [pin, 5, pout[:8]]
            tr.absorb([nonce])
        idx = [v & ((1 << (shape.logn + shape.logb)) - 1) for v in tr.squeeze(shape.n_queries)]
    except (KeyError, TypeError, IndexError) as e:
        raise ValueError("inner proof is malformed (%s)" % e)
    if idx != [int(q["index"]) for q in proof["queries"]]:
        raise ValueError("the query indices of an inner proof do not follow its transcript: no accepting witness")
    rec = tr.rec + ([pow_rec] if pow_rec else [])
    script = shape.transcript_perms()
    if [(r[1], r[2] is not None) for r in rec] != [(pm["n_in"], pm["out"]) for pm in script]:
        raise ValueError("inner proof does not have the shape the verifier AIR was built for")
    tp = []
    for inp, n, out in rec:
        tp += inp[:n]
        if out is not None:
            tp += out
    return [r[0] for r in rec], tp


def expected_publics(shape, proofs):
    """the Merkle part of the public inputs (the transcript part follows it: replay_transcript): roots of every inner proof, then
    the leaf index of every (slot, proof, tree), then the opened values of every (slot, proof, tree): slot g re-opens query
    g mod n_queries"""
    pubs = []
    for pr in proofs:
        names = {"trace": pr["roots"]["trace"], "quotient": pr["roots"]["quotient"]}
        if shape.W2:
            names["stage2"] = pr["roots"]["stage2"]
        for t, (name, _, _) in enumerate(shape.trees):
            root = pr["fri"]["roots"][int(name[3:])] if name.startswith("fri") else names[name]
            pubs += [int(v) for v in root]
    for g in range(shape.n_slots()):
        for pr in proofs:
            j = pr["queries"][g % shape.n_queries]["index"]
            for (_, _, depth) in shape.trees:
                pubs.append(int(j) & ((1 << depth) - 1))
    for g in range(shape.n_slots()):
        for pr in proofs:
            q = pr["queries"][g % shape.n_queries]
            for (name, w, _) in shape.trees:
                vals = _opening(q, name)["values"]
                if len(vals) != w:
                    raise ValueError("opening of %s has the wrong shape" % name)
                pubs += [int(v) for v in vals]
    return pubs


def prepare_proof(proof):
    """the query openings of an inner proof as arrays -- {"index": u64[nq], "values": [u64[nq][w] per tree], "paths": [u64[nq][depth][4]
    per tree]} in the order of Shape.trees -- so that the witness builder touches no Python list per opening.  The service
    prepares its own chunk proofs while the other chunks of the batch are still being proven (engine.py).  Raises ValueError for
    ragged or out-of-range data."""
    shape = Shape.of_proof(proof, 1)
    qs = proof["queries"]
    if len(qs) != shape.n_queries:
        raise ValueError("inner proof has the wrong number of queries")
    try:
        out = {"index": np.array([int(q["index"]) for q in qs], dtype=np.uint64), "values": [], "paths": [], "key": shape.key()[:8]}
        for (name, w, depth) in shape.trees:
            ops = [_opening(q, name) for q in qs]
            v = np.array([o["values"] for o in ops], dtype=np.uint64)
            pth = np.array([o["path"] for o in ops], dtype=np.uint64)
            if v.shape != (len(qs), w) or pth.shape != (len(qs), depth, 4):
                raise ValueError("opening of %s has the wrong shape" % name)
            out["values"].append(v)
            out["paths"].append(pth)
    except (OverflowError, TypeError, KeyError, IndexError) as e:
        raise ValueError("opening values are not field elements (%s)" % e)
    except ValueError as e:
        raise ValueError("opening of a tree has the wrong shape (%s)" % e)
    return out


_OPS_CACHE = {}


def _opening_table(shape):
    """one row per opening of the schedule, in block order: first block, absorb blocks, depth, tree, proof, query number -- fixed by
    the shape"""
    cached = _OPS_CACHE.get(shape.key())
    if cached is not None:
        return cached
    k, periods, pb = shape.layout()
    sched = shape.period_schedule()
    rows = []
    for per in range(periods):
        b = 0
        while b < pb:
            blk = sched[b]
            if blk["kind"] == "idle" or not blk["first"]:
                b += 1
                continue
            _, w, depth = shape.trees[blk["t"]]
            a = Shape.absorb_blocks(w)
            rows.append((per * pb + b, a, depth, blk["t"], blk["p"], (per * k + blk["sub"]) % shape.n_queries))
            b += a + depth
    tab = np.array(rows, dtype=np.int64).reshape(-1, 6)
    out = {"b0": tab[:, 0], "na": tab[:, 1], "nd": tab[:, 2], "t": tab[:, 3], "p": tab[:, 4], "q": tab[:, 5],
           "sel": {(p, t): np.nonzero((tab[:, 4] == p) & (tab[:, 3] == t))[0] for p in range(shape.n_proofs) for t in range(len(shape.trees))}}
    _OPS_CACHE[shape.key()] = out
    return out


def _publics_from_arrays(shape, proofs, prepared):
    """expected_publics() from prepared arrays (the same list, as one u64 array)"""
    parts = []
    for pr in proofs:
        names = {"trace": pr["roots"]["trace"], "quotient": pr["roots"]["quotient"]}
        if shape.W2:
            names["stage2"] = pr["roots"]["stage2"]
        for (name, _, _) in shape.trees:
            root = pr["fri"]["roots"][int(name[3:])] if name.startswith("fri") else names[name]
            parts.append(np.array([int(v) for v in root], dtype=np.uint64))
    g = np.arange(shape.n_slots()) % shape.n_queries
    masks = np.array([(1 << d) - 1 for (_, _, d) in shape.trees], dtype=np.uint64)
    parts.append(np.stack([pp["index"][g][:, None] & masks[None, :] for pp in prepared], axis=1).reshape(-1))     # [slot][proof][tree]
    parts.append(np.stack([np.concatenate([v[g] for v in pp["values"]], axis=1) for pp in prepared], axis=1).reshape(-1))
    return np.concatenate(parts)


def build_witness(shape, proofs, be, digest_words, prepared=None):
    """(trace u64[26][N], publics) for inner proof objects `proofs` (len = shape.n_proofs) of `shape`; the trace is a host array,
    or a device buffer of that shape when the backend assembles it in HBM (verifier_trace_device).
    be: backend with poseidon_perm_batch(states [B][12]), poseidon_trace(inputs [B][12]) -> (states [12][32 B], cubes
    [12][32 B]), poseidon_perm / poseidon_sponge_caps and publics_digest_gl for the transcripts; digest_words: the inner AIR's
    digest words (Air.digest_words(): the verifier of an inner proof knows its statement).  Raises ValueError when an opening
    does not hash to its root, or the transcript does not give the proof's indices / grinding -- there is no accepting witness
    for a proof that does not verify."""
    assert len(proofs) == shape.n_proofs
    for pr in proofs:
        assert Shape.of_proof(pr, shape.n_proofs).key() == shape.key(), "inner proofs of different shapes"
    if prepared is None:
        prepared = [prepare_proof(pr) for pr in proofs]
    k, periods, pb = shape.layout()
    nblk = pb * periods
    N = ROWS * nblk
    inputs = np.zeros((nblk, 12), dtype=np.uint64)
    dbit = np.zeros(nblk, dtype=np.uint64)
    idxv = np.zeros(nblk, dtype=np.uint64)
    # the openings as flat arrays (one row per opening: first block, absorb blocks, depth, index, padded values, path), then
    # level-synchronous hashing: absorb block j / tree level lv of EVERY opening is one batched permutation call
    max_w = max(8 * Shape.absorb_blocks(w) if w > 4 else 4 for (_, w, _) in shape.trees)
    max_d = max(d for (_, _, d) in shape.trees)
    ops = _opening_table(shape)
    b0, na, nd, tl, pl = ops["b0"], ops["na"], ops["nd"], ops["t"], ops["p"]
    vals = np.zeros((len(b0), max_w), dtype=np.uint64)
    paths = np.zeros((len(b0), max_d, 4), dtype=np.uint64)            # [openings][max_d][4]
    index = np.zeros(len(b0), dtype=np.uint64)
    for p, pp in enumerate(prepared):
        if pp["key"] != shape.key()[:8]:
            raise ValueError("inner proofs of different shapes")
        for t, (_, w, depth) in enumerate(shape.trees):
            sel = ops["sel"][(p, t)]
            q = ops["q"][sel]
            vals[sel, :w] = pp["values"][t][q]
            paths[sel, :depth] = pp["paths"][t][q]
            index[sel] = pp["index"][q] & np.uint64((1 << depth) - 1)
    nops = len(b0)
    cap = np.zeros((nops, 4), dtype=np.uint64)
    for j in range(int(na.max()) if nops else 0):
        sel = np.nonzero(na > j)[0]
        st = np.zeros((len(sel), 12), dtype=np.uint64)
        st[:, :8] = vals[sel, 8 * j:8 * j + 8]
        st[:, 8:] = cap[sel]
        inputs[b0[sel] + j] = st
        cap[sel] = be.poseidon_perm_batch(st)[:, :4]
    digest = np.where((na > 0)[:, None], cap, vals[:, :4])     # leaves of <= 4 values are not hashed (identity, zero padded)
    for lv in range(max_d):
        sel = np.nonzero(nd > lv)[0]
        if len(sel) == 0:
            break
        bit = (index[sel] >> np.uint64(lv)) & np.uint64(1)
        sib, cur = paths[sel, lv], digest[sel]
        st = np.zeros((len(sel), 12), dtype=np.uint64)
        st[:, :4] = np.where(bit[:, None] == 1, sib, cur)
        st[:, 4:8] = np.where(bit[:, None] == 1, cur, sib)
        blk = b0[sel] + na[sel] + lv
        inputs[blk] = st
        dbit[blk] = bit
        idxv[blk] = index[sel] & np.uint64((2 << lv) - 1)
        digest[sel] = be.poseidon_perm_batch(st)[:, :4]
    pub_parts = [_publics_from_arrays(shape, proofs, prepared)]
    L = len(shape.transcript_perms())
    for p, pr in enumerate(proofs):                 # the transcripts: consecutive blocks in the idle tail of the last period
        states, tp = replay_transcript(shape, pr, digest_words, be)
        blk0 = shape.transcript_block0() + p * L
        inputs[blk0:blk0 + L] = np.array(states, dtype=np.uint64)
        pub_parts.append(np.array(tp, dtype=np.uint64))
    pubs = np.concatenate(pub_parts)
    assert len(pubs) == shape.n_pub()
    T = len(shape.trees)
    want = pubs[:shape.n_proofs * T * 4].reshape(shape.n_proofs, T, 4)
    bad = np.nonzero((digest != want[pl, tl]).any(axis=1))[0]
    if len(bad):
        o = int(bad[0])
        raise ValueError("an opening of the %s tree of inner proof %d does not hash to its root: no accepting witness"
                         % (shape.trees[tl[o]][0], pl[o]))
    if hasattr(be, "verifier_trace_device"):      # GPU backend: the trace is assembled in HBM and stays there
        return be.verifier_trace_device(inputs, dbit, idxv), pubs
    states, cubes = be.poseidon_trace(inputs)
    trace = np.zeros((WIDTH, N), dtype=np.uint64)
    trace[S0:S0 + 12], trace[U0:U0 + 12] = states, cubes
    trace[COL_D] = np.repeat(dbit, ROWS)
    trace[COL_IDX] = np.repeat(idxv, ROWS)
    return trace, pubs


def aggregation_params(shape, n_queries=50, fri_logf=3, fri_final_log=5, pow_bits=0, hash="gl"):
    """STARK parameters of the proof over the verifier AIR: blow-up 4 (degree-4 constraints), so 2 bits per query"""
    from .prover import StarkParams
    logn = shape.logn_trace()
    return StarkParams(logn, 2, fri_logf, min(fri_final_log, logn - 1), n_queries, pow_bits, hash=hash)
