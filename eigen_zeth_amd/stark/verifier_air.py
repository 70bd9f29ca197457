"""The STARK-verifier AIR: verifying chunk proofs at their queries -- Merkle paths, transcripts AND the field arithmetic -- as constraints.

(Round 4: the opened values are PRIVATE.  The DEEP quotient, every FRI fold and the evaluation points of a query are constraints over the
values the permutation blocks hash; see "Arithmetic" below.  The paragraphs up to there describe the hashing part, which is unchanged except
that the opened values are no longer public inputs.)

Serves GenAggregatedProof (proto/prover/v1/prover.proto:115-126; client src/prover/provider.rs:422-451) and the final STARK
of GenFinalProof (prover.proto:130-148; provider.rs:472-503): "aggregate(p1, p2)" is a STARK whose witness is the
verification trace of the inner proofs' query openings -- every Poseidon permutation a verifier runs to check that the
opened rows of the trace / stage-2 / quotient / FRI-layer commitments hash up to the roots the proofs name -- and whose
public inputs are those roots and the query indices.  The reference holds no prover arithmetic, so the
construction is this repo's own (parity unpinned); it follows the public recursive-STARK recipe.

What the AIR proves (public inputs: per inner proof and tree the root; per query slot / proof / tree the leaf index; the transcripts;
a few constants per inner proof that its header determines):
    for every query slot, inner proof and committed tree there are leaf values and an authentication path such that linear_hash(values)
    hashed up the path, with the direction bits of the PUBLIC index, equals the PUBLIC root -- and those values satisfy the verifier's
    arithmetic at that query: they give the DEEP quotient at the query point, every FRI layer's opened coset holds the value the layer before
    it claims and folds to the value the next layer holds, the last fold is the public final-layer value.
What is left of verifying an inner proof needs no opening: reading the Fiat-Shamir transcript (which yields challenges and indices), the
out-of-domain constraint identity, the low-degree test of the final layer -- the checker of an aggregated proof does exactly that on the
inner proofs' HEADERS (oracle/aggregate_verify.py: stark_verify.verify(header_only=True)) and requires the outer proof's public inputs to be
what the headers dictate.  So an accepted aggregated proof means both inner proofs verify, and it carries none of their openings.

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
trace wraps to block 0.

Arithmetic (21 more columns; constraints stay at degree 4).  Per query slot and inner proof the blocks of the committed trees follow one
another -- trace, [stage 2], quotient, FRI layer 0, 1, ... -- each as its leaf (absorb blocks) and then its path (node blocks).  Next to them:
  HR[8]   the 8 rate values of the current absorb block (an unhashed leaf: its <= 4 values), constant over the block, bound to the state at row 0;
  X       the evaluation point spelled by the direction bits of the current path: X' = X (1 + d (c_l - 1)), c_l = w^(2^l) a fixed column --
          the quotient tree's path gives x = shift w_M^index, the path of FRI layer l its coset base point x_l = shift_l w_(lg)^row;
  XQN, XQ the point the NEXT layer is queried at (x_l^(2^f), accumulated along the same bits) and the latched copy of the previous one;
  ACA     phase A (trace / stage-2 / quotient leaves): Horner accumulator H <- H g^n + sum_i v_i g^(n-1-i) over the opened values, g = 1/gamma,
          so that sum_k gamma^k v_k = gamma^(K-1) H (the public powers of g sit in fixed columns: 8 values per row of ONE absorb block);
  ACB     the snapshot of H after the trace + stage-2 values (the second DEEP sum runs over those only).
  At the end of the quotient path:  F(x) = (gamma^(Wall-1) H - E_z) / (x - zeta) + (gamma^(Wall+Wt-1) H_B - E_zw) / (x - zeta w), with the two inverses as
  witnesses (in HR of that block) and E_z = sum_k gamma^k ev_k(zeta), E_zw likewise PUBLIC (O(columns) work per inner proof for whoever checks the
  public inputs, like the out-of-domain identity).  ACA <- -F(x).
  FRI layer l (leaf = the 2^f points of one coset, 3 x 2^f values in ceil(3 2^f / 8) absorb blocks): on rows j < 2^f of each absorb block
  D_j = (1 / 2^f) sum_k w_f^(-jk) (the block's values, as F_p^3 components) -- a fixed linear map, 24 fixed columns -- is accumulated twice:
      ACA += tau^j D_j        tau = x_q / x_l   (base field): the interpolant of the coset at the QUERY point = the opened value the previous
                                                  layer (or the DEEP quotient) claims -- ACA must be 0 when the layer's path ends;
      ACB += xi^j beta^j D_j  xi = 1 / x_l: the fold at beta (public powers of beta in fixed columns), handed on as -ACA of the next layer;
  xi and tau are witnesses held over the layer's blocks and checked against X when its path ends (xi X = 1, tau X = XQ); after the last layer
  ACB must be the PUBLIC value of the final layer at the query's position (a lookup for the checker, no arithmetic).
So the checker of an aggregated proof runs, per inner proof, the transcript READ, the out-of-domain identity, the low-degree test of the final layer
and O(columns) sums -- nothing per query but the index bookkeeping; the inner proofs' opened values are not part of the aggregated proof."""
from __future__ import annotations

import numpy as np

from . import air as A
from .air import Col, Fixed, FixedCol, Pub, Const

P = A.P
ROWS = 32          # rows per permutation block
N_ROUNDS, N_FULL_HALF = 30, 4
S0, U0, COL_D, COL_IDX = 0, 12, 24, 25
HR0, ACA0, ACB0, COL_XI, COL_TPX, COL_TAU, COL_TPT, COL_X, COL_XQ, COL_XQN, WIDTH = 26, 34, 37, 40, 41, 42, 43, 44, 45, 46, 47
Q_PIECES = 3       # degree-4 constraints: the quotient of a proof over this AIR is committed in 3 pieces (9 base columns), blow-up 4
ROOT32_DEFAULT, SHIFT_DEFAULT = 1753635133440165772, 49
# the arithmetic section of the public inputs, per inner proof: g^1..g^8 (g = 1 / gamma), then six F_p^3 constants, then beta_l^j
AP_G, AP_CA, AP_CB, AP_EZA, AP_EZB, AP_ZETA, AP_ZETAW, AP_BETA = 0, 24, 27, 30, 33, 36, 39, 42


class Shape:
    """what the Merkle part of a verifier needs to know about the inner proofs (all inner proofs of one aggregation share it)"""

    def __init__(self, logn, logb, W, W2, Wq, n_queries, fri_logf, fri_final_log, n_proofs=2, n_pub_inner=0, pow_bits=0, root32=None, shift=None):
        """n_pub_inner / pow_bits: number of public inputs and grinding bits of an inner proof (they shape its transcript); root32 / shift:
        the evaluation domain of the inner proofs (2^32-th root of unity, coset shift: constants of the arithmetic constraints)"""
        self.logn, self.logb, self.W, self.W2, self.Wq = logn, logb, W, W2, Wq
        self.n_queries, self.fri_logf, self.fri_final_log, self.n_proofs = n_queries, fri_logf, fri_final_log, n_proofs
        self.n_pub_inner, self.pow_bits = n_pub_inner, pow_bits
        self.root32 = ROOT32_DEFAULT if root32 is None else int(root32)
        self.shift = SHIFT_DEFAULT if shift is None else int(shift)
        logm = logn + logb
        self.final_log = logm
        self.trees = [("trace", W, logm)]
        if W2:
            self.trees.append(("stage2", W2, logm))
        self.trees.append(("quotient", Wq, logm))
        self.t_quot = len(self.trees) - 1          # trees 0 .. t_quot carry the values of the DEEP sums, the FRI layers follow
        cur, stop, li = logm, fri_final_log + logb, 0
        self.fri = []                              # (log size of the committed layer, log fold factor)
        while cur > stop:
            f = min(fri_logf, cur - stop)
            self.trees.append(("fri%d" % li, 3 << f, cur - f))
            self.fri.append((cur, f))
            cur -= f
            li += 1
        self.final_log = cur                       # log2 of the FRI layer sent in clear
        self.n_fri = li
        assert all(d >= 1 for (_, _, d) in self.trees)
        assert li >= 1 and logm >= 2, "the arithmetic constraints need at least one committed FRI layer"

    @staticmethod
    def of_proof(proof, n_proofs=2):
        """shape of a proof object (dict as stark/prover.py writes it)"""
        pr = proof["params"]
        q0 = proof["queries"][0]
        return Shape(pr["logn"], pr["logb"], len(q0["trace"]["values"]), len(q0["stage2"]["values"]) if "stage2" in q0 else 0,
                     len(q0["quotient"]["values"]), pr["n_queries"], pr["fri_logf"], pr["fri_final_log"], n_proofs,
                     len(proof["publics"]), pr["pow_bits"], int(proof["root32"]), int(proof["shift"]))

    KEY_NAMES = ("logn", "logb", "W", "W2", "Wq", "n_queries", "fri_logf", "fri_final_log", "n_proofs", "n_pub_inner", "pow_bits", "root32", "shift")

    def key(self):
        return (self.logn, self.logb, self.W, self.W2, self.Wq, self.n_queries, self.fri_logf, self.fri_final_log, self.n_proofs,
                self.n_pub_inner, self.pow_bits, self.root32, self.shift)

    def to_dict(self):
        return dict(zip(self.KEY_NAMES, self.key()))

    @staticmethod
    def from_dict(d):
        """shape named by an aggregated proof (untrusted text: bounded before anything is sized by it)"""
        v = [d[k] for k in Shape.KEY_NAMES]
        lim = dict(zip(Shape.KEY_NAMES, (30, 8, 4096, 4096, 64, 4096, 4, 16, 64, 1 << 24, 64, P - 1, P - 1)))
        if not all(isinstance(x, int) and not isinstance(x, bool) and 0 <= x <= lim[k] for k, x in zip(Shape.KEY_NAMES, v)):
            raise ValueError("shape out of range")
        if v[0] < 1 or v[1] < 1 or v[2] < 1 or v[4] < 1 or v[5] < 1 or v[6] < 1 or v[8] < 1 or v[0] + v[1] > 32 or v[11] < 2 or v[12] < 1:
            raise ValueError("shape out of range")
        if v[0] + v[1] <= v[7] + v[1] or v[0] + v[1] < 2:
            raise ValueError("shape without a committed FRI layer")
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
        """roots of every inner proof, then the leaf index of every (slot, proof, tree) -- the opened values are private"""
        T = len(self.trees)
        return self.n_proofs * T * 4 + self.n_slots() * self.n_proofs * T

    # ---- the arithmetic section: per inner proof, then one final-layer value per (slot, proof)
    def arith_pubs_per_proof(self):
        return AP_BETA + sum(3 * ((1 << f) - 1) for (_, f) in self.fri)

    def arith_base(self):
        return self.merkle_pubs() + self.n_proofs * self.transcript_pubs_per_proof()

    def pub_arith(self, p, off):
        return self.arith_base() + p * self.arith_pubs_per_proof() + off

    def ap_beta(self, li, j, c):
        """offset (within a proof's arithmetic section) of component c of beta_li^j, 1 <= j < 2^f"""
        return AP_BETA + sum(3 * ((1 << f) - 1) for (_, f) in self.fri[:li]) + 3 * (j - 1) + c

    def pub_finv(self, slot, p, c):
        """component c of the final-layer value of proof p at the position of query slot `slot`"""
        return self.arith_base() + self.n_proofs * self.arith_pubs_per_proof() + (slot * self.n_proofs + p) * 3 + c

    def deep_blocks(self, t):
        """the leaf blocks of tree t <= t_quot as (number of values n in the block) -- absorb blocks, or ONE entry for an unhashed leaf"""
        w = self.trees[t][1]
        a = self.absorb_blocks(w)
        return [min(8, w - 8 * j) for j in range(a)] if a else [w]

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
        return self.arith_base() + self.n_proofs * self.arith_pubs_per_proof() + self.n_slots() * self.n_proofs * 3

    def pub_root(self, p, t, i):
        return (p * len(self.trees) + t) * 4 + i

    def pub_index(self, slot, p, t):
        T = len(self.trees)
        return self.n_proofs * T * 4 + (slot * self.n_proofs + p) * T + t

    def values_per_query(self):
        return sum(w for (_, w, _) in self.trees)

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


def _root_of_unity(shape, logk):
    """the primitive 2^logk-th root of unity of the inner proofs' domain"""
    return pow(shape.root32, 1 << (32 - logk), P)


def tree_points(shape, t):
    """for tree t >= t_quot: (x0, [c_l], xq0, [cq_l]) -- the point its path spells is x0 prod_l c_l^bit_l (the quotient tree: the query
    point x = shift w_M^index; FRI layer li: the base point of the opened coset, shift_li w_lg^row) and the point the NEXT layer is queried
    at, xq0 prod_l cq_l^bit_l (= the same point for the quotient tree, its 2^f-th power for a FRI layer)"""
    depth = shape.trees[t][2]
    if t == shape.t_quot:
        lg = shape.logn + shape.logb
        cs = [_root_of_unity(shape, lg - l) for l in range(depth)]
        return shape.shift, cs, shape.shift, cs
    li = t - shape.t_quot - 1
    lg, f = shape.fri[li]
    sh = pow(shape.shift, 1 << sum(ff for (_, ff) in shape.fri[:li]), P)
    cs = [_root_of_unity(shape, lg - l) for l in range(depth)]
    return sh, cs, pow(sh, 1 << f, P), [pow(c, 1 << f, P) for c in cs]


def fold_coefficients(shape, li, blk):
    """FRI layer li, absorb block blk: {(row j, component c, rate position k): coefficient} of D_j = (1/2^f) sum_k' w_f^(-j k') pts[k'] restricted to
    the values this block holds -- flattened leaf value v = 8 blk + k is component v div 2^f of point v mod 2^f"""
    _, f = shape.fri[li]
    n, out = 1 << f, {}
    winv = pow(_root_of_unity(shape, f), P - 2, P)
    ninv = pow(n, P - 2, P)
    for k in range(8):
        v = 8 * blk + k
        if v >= 3 * n:
            break
        c, kp = divmod(v, n)
        for j in range(n):
            out[(j, c, k)] = ninv * pow(winv, (j * kp) % n, P) % P
    return out


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
    T, TQ = len(shape.trees), shape.t_quot

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
    # arithmetic selectors (all periodic with the schedule)
    hupd, intree, xinit, clm1, xqinit, clqm1, powrow, abs0f, lka, lkb, snapsel, qfin, frifin, lastfin = ([] for _ in range(14))
    fcoef = [[[] for _ in range(8)] for _ in range(3)]                    # F[c][k]: the inverse-DFT rows of the folds
    gp = [[] for _ in range(3)]                                           # GP: g^n on H-update links; beta^j on the fold rows
    gcol = [[[] for _ in range(3)] for _ in range(8)]                     # G[i]: g^(n-1-i) on H-update links; six constants on QFIN rows

    def e3_pub_or_one(dst, row, p, off, one):
        """F_p^3 entry: the constant 1 or three public inputs of proof p's arithmetic section"""
        if one:
            dst[0].append((row, 1))
        else:
            for c in range(3):
                dst[c].append((row, Pub(shape.pub_arith(p, off + c))))
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
        # ---- arithmetic: what the link row of block b does
        tree_end = blk["kind"] == "node" and blk["last"]
        if blk["kind"] != "idle" and not tree_end:
            intree.append((row, 1))
        t = blk.get("t", -1)
        is_snap = tree_end and t == TQ - 1
        is_qfin = tree_end and t == TQ
        is_frifin = tree_end and t > TQ
        if is_snap:
            snapsel.append((row, 1))
        if is_qfin:
            qfin.append((row, 1))
            for i, off in enumerate((AP_CA, AP_CB, AP_EZA, AP_EZB, AP_ZETA, AP_ZETAW)):
                e3_pub_or_one(gcol[i], row, blk["p"], off, False)
        if is_frifin:
            frifin.append((row, 1))
            if t == T - 1:
                lastfin.append((row, 1))
        nt = nxt.get("t", -1)
        nxt_tree_start = nxt["kind"] != "idle" and (nxt["j"] == 0 if nxt["kind"] == "absorb" else nxt["first"])
        nxt_leaf = nxt["kind"] != "idle" and nt <= TQ and (nxt["kind"] == "absorb" or nxt["first"])     # a block that holds DEEP values
        if nxt_leaf:
            hupd.append((row, 1))
            n = shape.deep_blocks(nt)[nxt["j"] if nxt["kind"] == "absorb" else 0]
            if not (nt == 0 and nxt_tree_start):                          # the first leaf block of a group starts H afresh: GP = 0
                e3_pub_or_one(gp, row, nxt["p"], AP_G + 3 * (n - 1), False)
            for i in range(n):
                e3_pub_or_one(gcol[i], row, nxt["p"], AP_G + 3 * (n - 2 - i), n - 1 - i == 0)
        if nxt_tree_start and nt >= TQ:
            x0, _, xq0, _ = tree_points(shape, nt)
            xinit.append((row, x0))
            xqinit.append((row, xq0))
        if nxt["kind"] == "node" and nt >= TQ:
            _, cs_, _, cqs_ = tree_points(shape, nt)
            if cs_[nxt["level"]] != 1:
                clm1.append((row, (cs_[nxt["level"]] - 1) % P))
            if cqs_[nxt["level"]] != 1:
                clqm1.append((row, (cqs_[nxt["level"]] - 1) % P))
        if blk["kind"] != "idle":
            if not (nxt_leaf or is_qfin or is_frifin):
                lka.append((row, 1))
            if not (is_snap or is_qfin or is_frifin):
                lkb.append((row, 1))
        # ---- arithmetic: the fold rows inside an absorb block of a FRI layer
        if blk["kind"] == "absorb" and t > TQ:
            li = t - TQ - 1
            R = 1 << shape.fri[li][1]
            r0 = b * ROWS
            abs0f.append((r0, 1))
            for j in range(R - 1):
                powrow.append((r0 + j, 1))
            for j in range(R):
                e3_pub_or_one(gp, r0 + j, blk["p"], shape.ap_beta(li, j, 0) if j else 0, j == 0)
            for (j, c, kk), v in fold_coefficients(shape, li, blk["j"]).items():
                fcoef[c][kk].append((r0 + j, v))
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
    # row 0 of a leaf block: absorb block j holds values 8j .. 8j+7 of the leaf in the rate, the first node block of an unhashed leaf
    # (width <= 4) holds it as the left or right child -- PRIVATE: the HR registers copy them, nothing names them
    abs0, id0 = [], []
    leaf = [[] for _ in range(8)]
    for b, blk in enumerate(sched):
        if blk["kind"] == "absorb":
            abs0.append((b * ROWS, 1))
        elif blk["kind"] == "node" and blk["first"]:
            id0.append((b * ROWS, 1))
    # the public final-layer value the last fold of (slot, proof) must give sits in LEAF[0..2] on that group's last row
    for per in range(periods):
        for (row, sub, p, t) in final_rows:
            if t == T - 1:
                for c in range(3):
                    leaf[c].append((per * (ROWS * pb) + row, Pub(shape.pub_finv(per * k + sub, p, c))))
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
    nf = 2 + len(fc)

    def add(entries):
        nonlocal nf
        fc.append(FixedCol(lp, entries))
        nf += 1
        return Fixed(nf - 1)
    HUPD, INTREE, XINIT, CLM1, XQINIT, CLQM1, POWROW, ABS0F, LKA, LKB, SNAPSEL, QFIN, FRIFIN, LASTFIN = \
        [add(e) for e in (hupd, intree, xinit, clm1, xqinit, clqm1, powrow, abs0f, lka, lkb, snapsel, qfin, frifin, lastfin)]
    FC = [[add(fcoef[c][kk]) for kk in range(8)] for c in range(3)]
    GP = [add(gp[c]) for c in range(3)]
    G = [[add(gcol[i][c]) for c in range(3)] for i in range(8)]

    # ---- constraints: hashing
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
    INBLK = ACT + CPY                                                     # rows 0 .. 30 of every block
    cs.append(idx_n - (INBLK + L_CHAIN + L_NODE) * idx - WT * d_n)        # index: kept inside a block and an opening, + 2^level * bit
    cs += [L_FINAL * s[i] - ROOT[i] for i in range(4)]                    # the top of the path is the public root
    cs.append(L_FINAL * idx - IDXV)                                       # the direction bits spell the public index
    # the absorbed blocks and the read rates of the transcripts are the public ones
    cs += [TABS * (s[i] - LEAF[i]) for i in range(8)]

    # ---- constraints: arithmetic on the (private) opened values
    hr = [Col(HR0 + i) for i in range(8)]
    hrn = [Col(HR0 + i, True) for i in range(8)]
    aca, acan = [Col(ACA0 + c) for c in range(3)], [Col(ACA0 + c, True) for c in range(3)]
    acb, acbn = [Col(ACB0 + c) for c in range(3)], [Col(ACB0 + c, True) for c in range(3)]
    xi, xin, tpx, tpxn = Col(COL_XI), Col(COL_XI, True), Col(COL_TPX), Col(COL_TPX, True)
    tau, taun, tpt, tptn = Col(COL_TAU), Col(COL_TAU, True), Col(COL_TPT), Col(COL_TPT, True)
    X, Xn, XQ, XQn, XQN, XQNn = Col(COL_X), Col(COL_X, True), Col(COL_XQ), Col(COL_XQ, True), Col(COL_XQN), Col(COL_XQN, True)
    # the registers: constant inside a block, the leaf's values at its row 0
    cs += [INBLK * (hrn[i] - hr[i]) for i in range(8)]
    cs += [ABS0 * (hr[i] - s[i]) + ID0 * (hr[i] - (s[i] + d_c * (s[4 + i] - s[i]))) for i in range(4)]
    cs += [ABS0 * (hr[i] - s[i]) + ID0 * hr[i] for i in range(4, 8)]
    # evaluation points along the paths
    KEEP = INBLK + INTREE
    cs.append(Xn - (KEEP * X + XINIT) * (1 + d_n * CLM1))
    cs.append(XQNn - (KEEP * XQN + XQINIT) * (1 + d_n * CLQM1))
    cs.append(XQn - KEEP * XQ - L_FINAL * XQN)                             # the query point of the next layer is latched where a path ends
    # xi = 1 / x_l and tau = x_q / x_l are held over a layer's blocks; their powers run down the fold rows
    cs.append(KEEP * (xin - xi))
    cs.append(KEEP * (taun - tau))
    cs.append(ABS0F * (tpx - 1))
    cs.append(ABS0F * (tpt - 1))
    cs.append(POWROW * (tpxn - tpx * xi))
    cs.append(POWROW * (tptn - tpt * tau))
    cs.append(FRIFIN * (xi * X - 1))
    cs.append(FRIFIN * (tau * X - XQ))
    # D_j of the fold rows (zero elsewhere: the coefficient columns are)
    D = []
    for c in range(3):
        acc = None
        for kk in range(8):
            if fcoef[c][kk]:
                term = FC[c][kk] * hr[kk]
                acc = term if acc is None else acc + term
        D.append(acc if acc is not None else Const(0))
    i1, i2 = hr[0:3], hr[3:6]
    CA_, CB_, EZA_, EZB_, ZETA_, ZETAW_ = G[0], G[1], G[2], G[3], G[4], G[5]
    horner = A.e3x_mul(aca, GP)                                            # H g^n ...
    for c in range(3):
        acc = horner[c]
        for i in range(8):
            acc = acc + hrn[i] * G[i][c]                                   # ... + sum_i v_i g^(n-1-i): the NEXT row's registers hold the block's values
        horner[c] = acc
    ta = A.e3x_mul(CA_, aca)
    tb = A.e3x_mul(CB_, acb)
    fx = A.e3x_mul([ta[c] - EZA_[c] for c in range(3)], i1)
    fx2 = A.e3x_mul([tb[c] - EZB_[c] for c in range(3)], i2)
    bd = A.e3x_mul(GP, D)                                                  # beta^j D_j
    for c in range(3):
        cs.append(acan[c] - (INBLK + LKA) * aca[c] - HUPD * horner[c] - tpt * D[c] + QFIN * (fx[c] + fx2[c]) + (FRIFIN - LASTFIN) * acb[c])
        cs.append(acbn[c] - (INBLK + LKB) * acb[c] - SNAPSEL * aca[c] - tpx * bd[c])
    cs += [FRIFIN * aca[c] for c in range(3)]                             # the interpolant of the coset at the query point IS the claimed value
    cs += [LASTFIN * (acb[c] - LEAF[c]) for c in range(3)]                # the last fold is the public final-layer value
    one = [Const(1), Const(0), Const(0)]
    inv1 = A.e3x_mul([X - ZETA_[0], Const(0) - ZETA_[1], Const(0) - ZETA_[2]], i1)
    inv2 = A.e3x_mul([X - ZETAW_[0], Const(0) - ZETAW_[1], Const(0) - ZETAW_[2]], i2)
    cs += [QFIN * (inv1[c] - one[c]) for c in range(3)]
    cs += [QFIN * (inv2[c] - one[c]) for c in range(3)]
    air = A.Air("mverify", WIDTH, shape.n_pub(), cs, trace_kind=None, fixed_cols=fc)
    air.shape = shape
    assert A.quotient_chunks(air) <= Q_PIECES
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
        chal = {"zeta": tr.squeeze(3), "betas": []}
        for r in proof["evals"]["z"] + proof["evals"]["zw"]:
            tr.absorb(r)
        chal["gamma"] = tr.squeeze(3)
        for root in proof["fri"]["roots"]:
            tr.absorb(root)
            chal["betas"].append(tr.squeeze(3))
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
    if "queries" in proof and idx != [int(q["index"]) for q in proof["queries"]]:      # a HEADER (publics_from_headers) carries no openings
        raise ValueError("the query indices of an inner proof do not follow its transcript: no accepting witness")
    chal["indices"] = idx
    rec = tr.rec + ([pow_rec] if pow_rec else [])
    script = shape.transcript_perms()
    if [(r[1], r[2] is not None) for r in rec] != [(pm["n_in"], pm["out"]) for pm in script]:
        raise ValueError("inner proof does not have the shape the verifier AIR was built for")
    tp = []
    for inp, n, out in rec:
        tp += inp[:n]
        if out is not None:
            tp += out
    return [r[0] for r in rec], tp, chal


def transcript_stream(shape, proof, digest_words):
    """what an inner proof's transcript absorbs, in protocol order, as one u64 array -- the form the native witness builder takes
    (zp_recursion_witness; layout in include/zeth_prover.h): AIR digest words | public inputs | roots of trace, [stage2], quotient |
    evaluations at zeta | at zeta w | FRI roots | final layer (3 planes) | grinding nonce (if any)"""
    try:
        out = [int(v) for v in digest_words] + [int(v) for v in proof["publics"]] + [int(v) for v in proof["roots"]["trace"]]
        if shape.W2:
            out += [int(v) for v in proof["roots"]["stage2"]]
        out += [int(v) for v in proof["roots"]["quotient"]]
        for r in proof["evals"]["z"] + proof["evals"]["zw"]:
            if len(r) != 3:
                raise ValueError("inner proof is malformed (evaluation)")
            out += [int(v) for v in r]
        for root in proof["fri"]["roots"]:
            out += [int(v) for v in root]
        for c in range(3):
            out += [int(v) for v in proof["fri"]["final"][c]]
        if shape.pow_bits:
            nonce = int(proof.get("pow_nonce", -1))
            if not 0 <= nonce < P:
                raise ValueError("grinding nonce missing")
            out.append(nonce)
        Wt = shape.W + shape.W2
        want = 4 + shape.n_pub_inner + 4 * len(shape.trees) + 3 * (Wt + shape.Wq) + 3 * Wt + (3 << shape.final_log) + (1 if shape.pow_bits else 0)
        if len(out) != want or any(not 0 <= v < P for v in out):
            raise ValueError("inner proof does not have the shape the verifier AIR was built for")
        return np.array(out, dtype=np.uint64)
    except (KeyError, TypeError, IndexError, OverflowError) as e:
        raise ValueError("inner proof is malformed (%s)" % e)


def arith_publics(shape, proof, chal):
    """the arithmetic section of ONE inner proof's public inputs (layout: AP_*): powers of g = 1 / gamma, the two DEEP constants
    gamma^(Wall-1), gamma^(Wall+Wt-1), the public halves of the two DEEP sums E_z = sum_k gamma^k ev_k(zeta), E_zw = sum_k gamma^(Wall+k) ev_k(zeta w),
    zeta, zeta w, and beta_l^j for every layer.  O(columns) field operations -- whoever checks the public inputs redoes exactly this."""
    from . import field as F
    gamma, zeta = F.e3(chal["gamma"]), F.e3(chal["zeta"])
    g = F.e3_inv(gamma)
    out, cur = [], [1, 0, 0]
    for _ in range(8):
        cur = F.e3_mul(cur, g)
        out += cur
    Wt = shape.W + shape.W2
    Wall = Wt + shape.Wq
    ev_all, ev_next = proof["evals"]["z"], proof["evals"]["zw"]
    if len(ev_all) != Wall or len(ev_next) != Wt:
        raise ValueError("inner proof has the wrong number of out-of-domain evaluations")
    eza, ezb, gk = [0, 0, 0], [0, 0, 0], [1, 0, 0]
    pw = []
    for k in range(Wall + Wt):
        pw.append(gk)
        gk = F.e3_mul(gk, gamma)
    for k in range(Wall):
        eza = F.e3_add(eza, F.e3_mul(pw[k], F.e3(ev_all[k])))
    for k in range(Wt):
        ezb = F.e3_add(ezb, F.e3_mul(pw[Wall + k], F.e3(ev_next[k])))
    wN = pow(shape.root32, 1 << (32 - shape.logn), P)
    out += pw[Wall - 1] + pw[Wall + Wt - 1] + eza + ezb + zeta + F.e3_scale(zeta, wN)
    for (_, f), beta in zip(shape.fri, chal["betas"]):
        cur = [1, 0, 0]
        for _ in range((1 << f) - 1):
            cur = F.e3_mul(cur, F.e3(beta))
            out += cur
    assert len(out) == shape.arith_pubs_per_proof()
    return out


def final_values(shape, proof, indices):
    """the final-layer value at the position of every query index: [len(indices)][3]"""
    fin = proof["fri"]["final"]
    m = (1 << shape.final_log) - 1
    if len(fin) != 3 or any(len(pl) != m + 1 for pl in fin):
        raise ValueError("inner proof has a final layer of the wrong size")
    return [[int(fin[c][int(j) & m]) for c in range(3)] for j in indices]


def publics_from_headers(shape, headers, digest_words, be):
    """The public inputs a STARK over verifier_air(shape) MUST have when its inner proofs have these headers -- derived from the headers alone
    (no opening is read): roots; the leaf index of every (slot, proof, tree), from each header's own Fiat-Shamir transcript; the transcripts,
    permutation by permutation; the arithmetic constants; the final-layer value at every query's position.  The same list build_witness returns
    beside the trace, in the same order.  GenFinalProof compares the client's aggregation STARK's public inputs with it (service/engine.py): an
    honest aggregation of the headers' proofs has exactly these, anything else vouches for other proofs than the ones the text carries.
    Raises ValueError for a header whose grinding nonce fails or whose shape is not `shape`."""
    if len(headers) != shape.n_proofs:
        raise ValueError("wrong number of inner proofs")
    parts, tps, chals = [], [], []
    for h in headers:
        names = {"trace": h["roots"]["trace"], "quotient": h["roots"]["quotient"]}
        if shape.W2:
            names["stage2"] = h["roots"]["stage2"]
        for (name, _, _) in shape.trees:
            root = h["fri"]["roots"][int(name[3:])] if name.startswith("fri") else names[name]
            if len(root) != 4:
                raise ValueError("malformed root")
            parts += [int(v) for v in root]
        _, tp, chal = replay_transcript(shape, h, digest_words, be)
        tps.append(tp)
        chals.append(chal)
    nq = shape.n_queries
    for g in range(shape.n_slots()):
        for chal in chals:
            j = chal["indices"][g % nq]
            parts += [int(j) & ((1 << depth) - 1) for (_, _, depth) in shape.trees]
    for tp in tps:
        parts += [int(v) for v in tp]
    for h, chal in zip(headers, chals):
        parts += arith_publics(shape, h, chal)
    for g in range(shape.n_slots()):
        for h, chal in zip(headers, chals):
            parts += final_values(shape, h, [chal["indices"][g % nq]])[0]
    if len(parts) != shape.n_pub():
        raise ValueError("inner proofs do not have the shape the verifier AIR was built for")
    return parts


def expected_publics(shape, proofs):
    """the Merkle part of the public inputs (the transcript part follows it: replay_transcript): roots of every inner proof, then
    the leaf index of every (slot, proof, tree): slot g re-opens query g mod n_queries.  The opened values are private."""
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


def prepare_proof_text(text):
    """(proof object WITHOUT its query openings but with "queries": n placeholders, prepared arrays) from a proof TEXT: the openings -- 97 % of
    the text -- are parsed by the library (native.parse_proof_queries: csrc/proofparse.hip), the header by the JSON module.  The object keeps
    a list of {"index": j} per query (what the witness builder and the shape need besides the arrays).  None when the text is not in the
    provers' own grammar: the caller falls back to json.loads + prepare_proof (which reports what is wrong)."""
    import json
    from .. import native
    data = text.encode() if isinstance(text, str) else text
    got = native.parse_proof_queries(data)
    if got is None:
        return None
    qb, qe, arr = got
    try:
        obj = json.loads(data[:qb] + b"[]" + data[qe:])
    except ValueError:
        return None
    if not isinstance(obj, dict) or not isinstance(obj.get("params"), dict):
        return None
    return obj, arr


def prepared_from_arrays(proof, arr):
    """the dictionary prepare_proof() returns, from natively parsed arrays; proof["queries"] becomes the list of {"index"} the other code reads"""
    pr = proof["params"]
    T = len(arr["values"])
    W = arr["values"][0].shape[1]
    W2 = arr["values"][1].shape[1] if arr["has_stage2"] else 0
    Wq = arr["values"][1 + (1 if arr["has_stage2"] else 0)].shape[1]
    shape = Shape(pr["logn"], pr["logb"], W, W2, Wq, pr["n_queries"], pr["fri_logf"], pr["fri_final_log"], 1, len(proof["publics"]), pr["pow_bits"],
                  int(proof["root32"]), int(proof["shift"]))
    if len(arr["index"]) != shape.n_queries or T != len(shape.trees):
        raise ValueError("inner proof has the wrong number of queries or commitments")
    for t, (name, w, depth) in enumerate(shape.trees):
        if arr["values"][t].shape != (shape.n_queries, w) or arr["paths"][t].shape != (shape.n_queries, depth, 4):
            raise ValueError("opening of %s has the wrong shape" % name)
    proof["queries"] = _QueryStubs(arr)
    return {"index": arr["index"], "values": arr["values"], "paths": arr["paths"], "key": shape.key()[:8]}


class _QueryStubs(list):
    """proof["queries"] of a natively parsed proof: per query {"index": j} (+ the leaf widths Shape.of_proof reads from query 0)"""

    def __init__(self, arr):
        names = ["trace"] + (["stage2"] if arr["has_stage2"] else []) + ["quotient"]
        q0 = {"index": int(arr["index"][0])}
        for t, n in enumerate(names):
            q0[n] = {"values": [0] * arr["values"][t].shape[1]}
        super().__init__([q0] + [{"index": int(j)} for j in arr["index"][1:]])


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
    return np.concatenate(parts)


def build_witness(shape, proofs, be, digest_words, prepared=None, keep=None):
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
    for pp in prepared:
        if pp["key"] != shape.key()[:8]:
            raise ValueError("inner proofs of different shapes")
    if hasattr(be, "recursion_witness") and keep is None:
        return be.recursion_witness(shape, proofs, prepared, digest_words)      # the same witness through ONE library call (csrc/recursion.hip)
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
    chals = []
    for p, pr in enumerate(proofs):                 # the transcripts: consecutive blocks in the idle tail of the last period
        states, tp, chal = replay_transcript(shape, pr, digest_words, be)
        chals.append(chal)
        blk0 = shape.transcript_block0() + p * L
        inputs[blk0:blk0 + L] = np.array(states, dtype=np.uint64)
        pub_parts.append(np.array(tp, dtype=np.uint64))
    aps = [arith_publics(shape, pr, ch) for pr, ch in zip(proofs, chals)]
    pub_parts += [np.array(a, dtype=np.uint64) for a in aps]
    gq = np.arange(shape.n_slots()) % shape.n_queries
    fin = np.stack([np.array(final_values(shape, pr, pp["index"][gq]), dtype=np.uint64) for pr, pp in zip(proofs, prepared)], axis=1)
    pub_parts.append(fin.reshape(-1))                                 # [slot][proof][3]
    pubs = np.concatenate(pub_parts)
    assert len(pubs) == shape.n_pub()
    T = len(shape.trees)
    want = pubs[:shape.n_proofs * T * 4].reshape(shape.n_proofs, T, 4)
    bad = np.nonzero((digest != want[pl, tl]).any(axis=1))[0]
    if len(bad):
        o = int(bad[0])
        raise ValueError("an opening of the %s tree of inner proof %d does not hash to its root: no accepting witness"
                         % (shape.trees[tl[o]][0], pl[o]))
    # the arithmetic columns: DEEP quotient, folds and evaluation points of every query (raises when an inner proof's values are
    # inconsistent: a wrong DEEP value, a fold that does not give the next layer's value, a last fold that is not the final layer)
    if "blk_op" not in ops:
        ops["blk_op"] = block_openings(shape, ops)
    arith_in = {"vals": vals, "index": index, "dbit": dbit, "blk_op": ops["blk_op"], "aps": aps, "fin": fin}
    if keep is not None:
        keep["arith_in"] = arith_in                                  # tests: the inputs of the arithmetic builders
    if hasattr(be, "verifier_arith_columns"):
        arith = be.verifier_arith_columns(shape, arith_in)           # native: same columns (tests compare the two)
    else:
        arith = arith_columns(shape, arith_in)
    if hasattr(be, "verifier_trace_device"):      # GPU backend: the trace is assembled in HBM and stays there
        return be.verifier_trace_device(inputs, dbit, idxv, arith), pubs
    states, cubes = be.poseidon_trace(inputs)
    trace = np.zeros((WIDTH, N), dtype=np.uint64)
    trace[S0:S0 + 12], trace[U0:U0 + 12] = states, cubes
    trace[COL_D] = np.repeat(dbit, ROWS)
    trace[COL_IDX] = np.repeat(idxv, ROWS)
    trace[HR0:] = arith
    return trace, pubs


_DESC_CACHE = {}
ARITH_DESC_MAGIC = int.from_bytes(b"PZVARITH", "little")


def arith_descriptor(shape):
    """the schedule of the arithmetic columns as data for the native builder (zp_verifier_arith_host / zp_verifier_arith_trace; layout in
    csrc/recursion.hip): header (schedule sizes + the inner proofs' shape) | one word per block of a period | one record per committed tree |
    the inverse-DFT tables of the folds | the transcript script (one word per permutation of an inner proof's sponge)"""
    cached = _DESC_CACHE.get(shape.key())
    if cached is not None:
        return cached
    k, periods, pb = shape.layout()
    sched = shape.period_schedule()
    T, TQ = len(shape.trees), shape.t_quot
    max_w = max(8 * Shape.absorb_blocks(w) if w > 4 else 4 for (_, w, _) in shape.trees)
    KIND = {"idle": 0, "absorb": 1, "node": 2}
    blk_words = []
    for blk in sched:
        if blk["kind"] == "idle":
            blk_words.append(0)
            continue
        jl = blk["j"] if blk["kind"] == "absorb" else blk["level"]
        blk_words.append(KIND[blk["kind"]] | blk["t"] << 2 | jl << 8 | int(bool(blk["first"])) << 16 | int(bool(blk["last"])) << 17 | blk["p"] << 18 |
                         blk["sub"] << 34)
    tree_words, fold_words, n_fold = [], [], 0
    for t, (_, w, depth) in enumerate(shape.trees):
        rec = [w, depth, 0, 0, 0, 0, 0, 0] + [0] * 64
        if t >= TQ:
            x0, cs_, xq0, cqs_ = tree_points(shape, t)
            rec[2], rec[3] = x0, xq0
            rec[8:8 + depth] = cs_
            rec[40:40 + depth] = cqs_
        if t > TQ:
            li = t - TQ - 1
            lg, f = shape.fri[li]
            rec[4], rec[5], rec[6], rec[7] = lg, f, shape.ap_beta(li, 1, 0), n_fold
            for bj in range(Shape.absorb_blocks(w)):
                tab = [0] * (16 * 3 * 8)
                for (j, c, kk), v in fold_coefficients(shape, li, bj).items():
                    tab[(j * 3 + c) * 8 + kk] = v
                fold_words += tab
                n_fold += 1
        tree_words += rec
    n_open = shape.n_slots() * shape.n_proofs * T
    script = [pm["n_in"] | int(pm["out"]) << 8 | int(pm["first"]) << 9 | int(pm["pow"]) << 10 for pm in shape.transcript_perms()]
    hdr = [ARITH_DESC_MAGIC, pb, periods, k, shape.n_proofs, T, TQ, max_w, n_open, shape.arith_pubs_per_proof(), n_fold, shape.n_queries, len(script),
           shape.transcript_block0(), shape.W, shape.W2, shape.Wq, shape.final_log, shape.n_pub_inner, shape.pow_bits, shape.logn, shape.logb,
           shape.fri_logf, shape.fri_final_log, shape.root32, shape.shift]
    desc = np.array(hdr + blk_words + tree_words + fold_words + script, dtype=np.uint64)
    _DESC_CACHE[shape.key()] = desc
    return desc


def block_openings(shape, ops):
    """block -> opening (row of _opening_table), -1 for idle / transcript blocks"""
    k, periods, pb = shape.layout()
    blk_op = np.full(pb * periods, -1, dtype=np.int64)
    ends = ops["b0"] + ops["na"] + ops["nd"]
    for o in range(len(ops["b0"])):
        blk_op[ops["b0"][o]:ends[o]] = o
    return blk_op


def link_roles(shape, blk, nxt):
    """what the link row between schedule blocks blk -> nxt does to the arithmetic registers: the selector values the AIR puts on that row
    (verifier_air() lists the same conditions)"""
    TQ, T = shape.t_quot, len(shape.trees)
    tree_end = blk["kind"] == "node" and blk["last"]
    t, nt = blk.get("t", -1), nxt.get("t", -1)
    r = {"keep": blk["kind"] != "idle" and not tree_end, "tree_end": tree_end, "snap": tree_end and t == TQ - 1, "qfin": tree_end and t == TQ,
         "frifin": tree_end and t > TQ, "lastfin": tree_end and t == T - 1}
    r["tree_start"] = nxt["kind"] != "idle" and (nxt["j"] == 0 if nxt["kind"] == "absorb" else nxt["first"])
    r["leaf"] = nxt["kind"] != "idle" and nt <= TQ and (nxt["kind"] == "absorb" or nxt["first"])
    r["lka"] = blk["kind"] != "idle" and not (r["leaf"] or r["qfin"] or r["frifin"])
    r["lkb"] = blk["kind"] != "idle" and not (r["snap"] or r["qfin"] or r["frifin"])
    return r


def arith_columns(shape, a):
    """The 21 arithmetic columns u64[21][N] (HR, ACA, ACB, XI, TPX, TAU, TPT, X, XQ, XQN) -- the REFERENCE builder: it walks the blocks in
    schedule order and applies, link by link and fold row by fold row, exactly the transitions the constraints state (Python integers; the
    native builder behind be.verifier_arith_columns produces the same columns).  a: {"vals" [openings][max_w], "index" [openings], "dbit"
    [blocks], "blk_op" [blocks] = block_openings(), "aps" per-proof arithmetic publics, "fin" [slots][proofs][3]}."""
    from . import field as F
    k, periods, pb = shape.layout()
    nblk = pb * periods
    sched = shape.period_schedule()
    TQ, T = shape.t_quot, len(shape.trees)
    blk_op = a["blk_op"]                                # block -> opening
    vals, index, dbit = a["vals"], a["index"], a["dbit"]
    N = ROWS * nblk
    out = np.zeros((WIDTH - HR0, N), dtype=np.uint64)
    C = {"hr": 0, "aca": ACA0 - HR0, "acb": ACB0 - HR0, "xi": COL_XI - HR0, "tpx": COL_TPX - HR0, "tau": COL_TAU - HR0, "tpt": COL_TPT - HR0,
         "x": COL_X - HR0, "xq": COL_XQ - HR0, "xqn": COL_XQN - HR0}
    pts = {t: tree_points(shape, t) for t in range(TQ, T)}
    coefs = {}
    ap = [[int(v) for v in x] for x in a["aps"]]
    e3at = lambda p, off: ap[p][off:off + 3]
    # registers at the END (row 31) of the previous block; the walk starts behind an idle block (all zero)
    X = XQ = XQN = XI = TAU = 0
    ACA, ACB, HR = [0, 0, 0], [0, 0, 0], [0] * 8
    prev = sched[-1]
    for gb in range(nblk):
        per, b = divmod(gb, pb)
        blk = sched[b]
        lr = link_roles(shape, prev, blk)
        o = int(blk_op[gb])
        d = int(dbit[gb])
        t = blk.get("t", -1)
        p = blk.get("p", 0)
        # ---- the link: registers at row 0 of this block
        x0 = xq0 = cm1 = cqm1 = 0
        if t >= TQ:
            tx0, tcs, txq0, tcqs = pts[t]
            if lr["tree_start"]:
                x0, xq0 = tx0, txq0
            if blk["kind"] == "node":
                cm1, cqm1 = tcs[blk["level"]] - 1, tcqs[blk["level"]] - 1
        keep = 1 if lr["keep"] else 0
        nXQ = (keep * XQ + (XQN if lr["tree_end"] else 0)) % P
        X = (keep * X + x0) * (1 + d * cm1) % P
        XQN = (keep * XQN + xq0) * (1 + d * cqm1) % P
        XQ = nXQ
        # the registers of this block
        nHR = [0] * 8
        if blk["kind"] == "absorb":
            nHR = [int(v) for v in vals[o, 8 * blk["j"]:8 * blk["j"] + 8]]
            nHR += [0] * (8 - len(nHR))
        elif blk["kind"] == "node" and blk["first"]:
            w = shape.trees[t][1]
            nHR = [int(v) for v in vals[o, :4]] + [0] * 4
            assert w <= 4
        elif blk["kind"] == "node" and blk["last"] and t == TQ:
            zeta, zeta_w = e3at(p, AP_ZETA), e3at(p, AP_ZETAW)
            i1 = F.e3_inv(F.e3_sub([X, 0, 0], zeta))
            i2 = F.e3_inv(F.e3_sub([X, 0, 0], zeta_w))
            nHR = i1 + i2 + [0, 0]
        nACA = [ACA[c] if lr["lka"] else 0 for c in range(3)]
        nACB = [ACB[c] if lr["lkb"] else 0 for c in range(3)]
        if lr["leaf"]:                                # H <- H g^n + sum_i v_i g^(n-1-i)
            n = shape.deep_blocks(t)[blk["j"] if blk["kind"] == "absorb" else 0]
            first_of_group = t == 0 and lr["tree_start"]
            h = [0, 0, 0] if first_of_group else F.e3_mul(ACA, e3at(p, AP_G + 3 * (n - 1)))
            for i in range(n):
                gi = [1, 0, 0] if n - 1 - i == 0 else e3at(p, AP_G + 3 * (n - 2 - i))
                h = F.e3_add(h, F.e3_scale(gi, nHR[i]))
            nACA = h
        if lr["snap"]:
            nACB = list(ACA)
        if lr["qfin"]:                                # -F(x) from H, H_B and the two inverses (in the registers of the block that ends)
            pp = prev["p"]
            fa = F.e3_mul(F.e3_sub(F.e3_mul(e3at(pp, AP_CA), ACA), e3at(pp, AP_EZA)), HR[0:3])
            fb = F.e3_mul(F.e3_sub(F.e3_mul(e3at(pp, AP_CB), ACB), e3at(pp, AP_EZB)), HR[3:6])
            nACA = F.e3_sub([0, 0, 0], F.e3_add(fa, fb))
        if lr["frifin"]:
            if any(ACA):
                raise ValueError("the opened values of an inner proof are inconsistent (a FRI layer does not hold the value the layer before "
                                 "it claims): no accepting witness")
            if lr["lastfin"]:
                slot = (gb - 1) // pb * k + prev["sub"]
                if [int(v) for v in a["fin"][slot][prev["p"]]] != ACB:
                    raise ValueError("the last fold of an inner proof does not give its final layer: no accepting witness")
            else:
                nACA = F.e3_sub([0, 0, 0], ACB)
        # xi, tau: held inside a tree; chosen where a FRI layer starts
        if not keep:
            XI = TAU = 0
            if t > TQ and lr["tree_start"]:
                lg, f = shape.fri[t - TQ - 1]
                row = int(index[o])
                xl = pts[t][0] * pow(_root_of_unity(shape, lg), row, P) % P
                XI = pow(xl, P - 2, P)
                TAU = XQ * XI % P
        ACA, ACB, HR = nACA, nACB, nHR
        # ---- rows of this block
        r0 = gb * ROWS
        rows = slice(r0, r0 + ROWS)
        for i in range(8):
            out[C["hr"] + i, rows] = HR[i]
        out[C["xi"], rows], out[C["tau"], rows], out[C["x"], rows], out[C["xq"], rows], out[C["xqn"], rows] = XI, TAU, X, XQ, XQN
        if blk["kind"] == "absorb" and t > TQ:        # the fold rows
            li = t - TQ - 1
            R = 1 << shape.fri[li][1]
            key = (li, blk["j"])
            if key not in coefs:
                coefs[key] = fold_coefficients(shape, li, blk["j"])
            cf = coefs[key]
            tpx = tpt = 1
            for j in range(R):
                D = [0, 0, 0]
                for (jj, c, kk), v in cf.items():
                    if jj == j:
                        D[c] = (D[c] + v * HR[kk]) % P
                bj = [1, 0, 0] if j == 0 else e3at(p, shape.ap_beta(li, j, 0))
                for c in range(3):
                    out[C["aca"] + c, r0 + j], out[C["acb"] + c, r0 + j] = ACA[c], ACB[c]
                out[C["tpx"], r0 + j], out[C["tpt"], r0 + j] = tpx, tpt
                ACA = F.e3_add(ACA, F.e3_scale(D, tpt))
                ACB = F.e3_add(ACB, F.e3_scale(F.e3_mul(bj, D), tpx))
                if j < R - 1:
                    tpx, tpt = tpx * XI % P, tpt * TAU % P
            for c in range(3):
                out[C["aca"] + c, r0 + R:r0 + ROWS], out[C["acb"] + c, r0 + R:r0 + ROWS] = ACA[c], ACB[c]
            out[C["tpx"], r0 + R:r0 + ROWS], out[C["tpt"], r0 + R:r0 + ROWS] = tpx, tpt
        else:
            for c in range(3):
                out[C["aca"] + c, rows], out[C["acb"] + c, rows] = ACA[c], ACB[c]
        prev = blk
    return out


def aggregation_params(shape, n_queries=50, fri_logf=3, fri_final_log=5, pow_bits=0, hash="gl"):
    """STARK parameters of the proof over the verifier AIR: blow-up 4 (degree-4 constraints), so 2 bits per query"""
    from .prover import StarkParams
    logn = shape.logn_trace()
    return StarkParams(logn, 2, fri_logf, min(fri_final_log, logn - 1), n_queries, pow_bits, hash=hash)
