"""The circuit of the Groth16 wrap (GenFinalProof, proto/prover/v1/prover.proto:130-148; the proof it makes is what
src/settlement/ethereum/mod.rs:338-394 hands to EigenZkVM.verifyBatches): an R1CS over the BN254 scalar field that verifies ALL THE HASHING of
the final STARK's verifier -- the Merkle openings at its queries (stage A, round 4) AND the Fiat-Shamir transcript that dictates WHERE those
queries are (stage B-1, round 5).

Statement (one public input d):  there are, for every query q of the final STARK and every committed tree t (trace, quotient, FRI layers):
the query index j_q with its bits, the leaf's field elements (the opened Goldilocks values packed 56 to a sponge block of 16 elements:
csrc/poseidon_bn254.hip, oracle/naive.py:pack_leaf_block), and per tree level the 16 digests of the group on the path, such that
  * the leaf sponge (width-17 Poseidon-BN254, capacity = element 0) of the leaf elements gives the leaf digest,
  * at every level the digest computed so far sits at position (index >> 4 level) & 15 of the group -- the position selected by the BITS of
    j_q --, and the group hashes ([0, 16 children] -> element 0) to the digest of the next level, the last one being the tree's root,
  * THE INDICES ARE THE TRANSCRIPT'S (stage B-1): the sponge of stark/transcript.py TranscriptBN128 -- capacity chained through one gadget per
    absorbed block of 16 elements, in the prover's order: parameters + statement digest (+ the public inputs or their commitment), the trace
    root | the quotient root | the out-of-domain evaluations | one FRI root per layer | the final layer -- ends in rate elements whose
    64-bit words, by a full (canonical, < r) bit decomposition in the circuit, ARE the bits of j_0, j_1, ...: the same wires select the
    path positions above.  (A word >= 2^64 - 2^32 would have to be reduced mod p first; the circuit requires that its top 32 bits are not all
    set instead: an honest transcript misses that with probability 2^-32 per query and is re-randomised by the caller's retry.)
  * d = the root of a 16-ary Poseidon tree over the list  roots | aux | transcript data (parameter / digest elements, publics commitment,
    out-of-domain evaluations, final layer: everything the sponge absorbs besides the roots) | per query: j_q, then the leaf elements of
    every tree  (aux: one more field element the proof is bound to -- the aggregator address of the request).
So d commits to everything the rest of a verifier needs -- roots, evaluations, final layer, indices, opened values -- and the circuit vouches
that the opened values ARE the leaves under those roots at the positions the transcript of exactly this data dictates: an index cannot be
chosen any more (round 4's circuit left the indices free: its proof said nothing without a native transcript check).  What stays outside
(checked natively by whoever holds the final STARK: oracle/wrap_verify.py): recomputing d from the final STARK's data, the packing of values
into elements, and the field arithmetic (out-of-domain identity, DEEP, folds; the challenges alpha, zeta, gamma, beta are read off the same
sponge natively) -- stage B-2.

Size at the service's parameters (final STARK: 2^18 rows x 47 columns, blow-up 4, 50 queries, 7 trees): 50 x (7 leaf + 26 node permutations)
+ 23 transcript + 365 data-tree gadgets x 613 constraints + 53 k path glue + 13 k for the bit decompositions = 1.31 M constraints (2^21
domain), 1.33 M wires."""
from __future__ import annotations

import numpy as np

from . import r1cs as R1
from ..stark.prover import StarkParams, bn128_rows_per_leaf_log

R = R1.R
LEAF_BLOCK = 56
_CACHE = {}


def pack_leaf_block(vals):
    """one sponge block of a leaf: up to 56 Goldilocks values in 16 field elements -- element e holds values 3e, 3e+1, 3e+2 in bits 0..191 and,
    in bits 192..223, 32-bit half number e of values 48..55 (the convention of csrc/poseidon_bn254.hip's leaf kernels)"""
    vals = [int(v) for v in vals] + [0] * (LEAF_BLOCK - len(vals))
    out = []
    for e in range(16):
        x = vals[48 + (e >> 1)]
        half = (x >> 32) if (e & 1) else (x & 0xFFFFFFFF)
        out.append(vals[3 * e] + (vals[3 * e + 1] << 64) + (vals[3 * e + 2] << 128) + (half << 192))
    return out


class Layout:
    """the trees of a BN128-mode STARK as the circuit sees them: per tree (name, values per leaf, leaves, [group count per level])"""

    def __init__(self, params, W, Wq, n_pub=0, head=None):
        """head: the values the prover absorbs first (stark/prover.py: parameters, root of unity, shift, statement digest words, public-input
        count) -- known to whoever builds the circuit only as a COUNT (their values are transcript data, committed in d)"""
        assert params.hash == "bn128"
        self.params = params
        self.W, self.Wq, self.n_pub = W, Wq, n_pub
        self.n_head = 15 if head is None else len(head)
        self.logm = params.logn + params.logb
        M = 1 << self.logm
        gt, qg = bn128_rows_per_leaf_log(W, self.logm), bn128_rows_per_leaf_log(Wq, self.logm)
        self.trees = [("trace", W << gt, M >> gt), ("quotient", Wq << qg, M >> qg)]
        sched, _ = params.fri_schedule()
        for li, (lg, f) in enumerate(sched):
            self.trees.append(("fri%d" % li, 3 << f, 1 << (lg - f)))
        self.n_queries = params.n_queries

    def key(self):
        return (tuple(sorted(self.params.to_dict().items())), tuple(self.trees), self.W, self.Wq, self.n_pub, self.n_head)

    @staticmethod
    def of_air(air, params):
        """the layout of a BN128-mode proof of `air` (stark/air.py Air) under `params`: the widths come from the STATEMENT, not from a proof text"""
        from ..stark import air as AIR
        assert not air.stage2, "the final STARK's AIR has no stage-2 columns"
        return Layout(params, air.width, 3 * AIR.quotient_chunks(air), air.n_pub)

    def transcript_segments(self):
        """what the prover's sponge absorbs between two squeezes (stark/prover.py, BN128 mode: no stage 2, no grinding), as lists of
        (kind, count): "head" parameter / digest (/ inline public input) elements, "pubs" the publics commitment, ("root", t) the root of
        tree t, "evals" one element per out-of-domain evaluation, "final" the final layer's elements (three planes, each padded on its own)"""
        from ..stark.prover import PUBLICS_INLINE
        inline = self.n_pub <= PUBLICS_INLINE
        seg0 = [("head", -(-(self.n_head + (self.n_pub if inline else 0)) // 3))] + ([] if inline else [("pubs", 1)]) + [(("root", 0), 1)]
        segs = [seg0, [(("root", 1), 1)], [("evals", 2 * self.W + self.Wq)]]
        sched, final_log = self.params.fri_schedule()
        for li in range(len(sched)):
            segs.append([(("root", 2 + li), 1)])
        segs.append([("final", 3 * -(-(1 << final_log) // 3))])
        return segs

    def squeeze_perms(self):
        """permutations whose rate the query indices are read from: 48 values (16 elements x 3 words) each"""
        return -(-self.n_queries // 48)

    @staticmethod
    def levels(n_leaves):
        out, n = [], n_leaves
        while n > 1:
            out.append(n)
            n = (n + 15) // 16
        return out

    @staticmethod
    def blocks(width):
        return max(1, -(-width // LEAF_BLOCK))


R_BITS = [(R >> i) & 1 for i in range(254)]


def perm17(state):
    """the width-17 Poseidon-BN254 permutation on Python integers (tables: poseidon_constants.bn254_poseidon_params) -- for the reference
    assignment of small circuits and for openings records of proofs that were not made by the library's one-call prover"""
    if "p17" not in _CACHE:            # (the tables come out of the Grain LFSR: seconds to regenerate)
        from ..poseidon_constants import bn254_poseidon_params
        _CACHE["p17"] = bn254_poseidon_params(17)
    rc, mds, rp = _CACHE["p17"]
    st = [int(v) % R for v in state]
    for r in range(8 + rp):
        st = [(v + rc[r * 17 + i]) % R for i, v in enumerate(st)]
        if r < 4 or r >= 4 + rp:
            st = [pow(v, 5, R) for v in st]
        else:
            st[0] = pow(st[0], 5, R)
        st = [sum(mds[i][j] * st[j] for j in range(17)) % R for i in range(17)]
    return st


def head_values(air, params, root32, shift):
    """what a prover absorbs first (stark/prover.py): every parameter the verifier relies on, the domain, the statement's digest, the public-input count"""
    return [params.logn, params.logb, air.width, air.width2, params.fri_logf, params.fri_final_log, params.n_queries, params.pow_bits, int(root32), int(shift)] \
        + air.digest_words() + [air.n_pub]


class TranscriptLog:
    """the sponge of a BN128-mode final STARK replayed from the proof's own data (stark/transcript.py TranscriptBN128; the order is
    stark/prover.py's): blocks = every block of 16 elements the sponge absorbed, in order; rates = the 16 rate elements of the last absorbing
    permutation and of the squeeze-only permutations behind it (as many as the query indices need); indices = what they dictate."""

    def __init__(self, proof, layout, head, publics_digest=None, perm=perm17):
        P = 0xFFFFFFFF00000001
        from ..stark.prover import PUBLICS_INLINE
        pack3 = lambda vals: [int(vals[i]) % P + ((int(vals[i + 1]) % P) << 64 if i + 1 < len(vals) else 0) + ((int(vals[i + 2]) % P) << 128 if i + 2 < len(vals) else 0)
                              for i in range(0, len(vals), 3)]
        pubs = [int(v) for v in proof["publics"]]
        roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
        if len(head) != layout.n_head or len(pubs) != layout.n_pub:
            raise ValueError("final STARK does not have the shape the wrap circuit was built for")
        if len(pubs) <= PUBLICS_INLINE:
            seg0 = pack3(list(head) + pubs)
        else:
            if publics_digest is None:
                raise ValueError("the commitment to the public inputs is needed (a backend's publics_digest)")
            seg0 = pack3(list(head)) + [int(publics_digest(pubs)[0])]
        segs = [seg0 + [int(roots[0][0])], [int(roots[1][0])], [pack3(r)[0] for r in proof["evals"]["z"] + proof["evals"]["zw"]]]
        segs += [[int(r[0])] for r in roots[2:]]
        segs.append([e for pl in proof["fri"]["final"] for e in pack3(pl)])
        want = [sum(cnt for _, cnt in seg) for seg in layout.transcript_segments()]
        if [len(sg) for sg in segs] != want:
            raise ValueError("final STARK does not have the shape the wrap circuit was built for")
        self.data = [e for sg, spec in zip(segs, layout.transcript_segments()) for e, kind in zip(sg, [k for k, cnt in spec for _ in range(cnt)])
                     if not isinstance(kind, tuple)]      # everything absorbed besides the roots, in order
        self.blocks, self.caps, state = [], [], [0] * 17          # caps: the capacity element after every permutation, in order
        self.chal = []                                            # rate element 1 after every segment: what the challenge squeezed there is read from
        for sg in segs:
            for off in range(0, len(sg), 16):
                blk = sg[off:off + 16] + [0] * (16 - len(sg[off:off + 16]))
                self.blocks.append(blk)
                state = perm([state[0]] + blk)
                self.caps.append(state[0])
            self.chal.append(state[1])
        self.rates = [state[1:]]
        for _ in range(layout.squeeze_perms() - 1):
            state = perm(state)
            self.rates.append(state[1:])
            self.caps.append(state[0])
        vals = [((e >> (64 * k)) & 0xFFFFFFFFFFFFFFFF) % P for rate in self.rates for e in rate for k in range(3)]
        self.indices = [v & ((1 << layout.logm) - 1) for v in vals[:layout.n_queries]]


class WrapCircuit:
    """the circuit + where its caller-set wires are.  wires: Z (value 0), per tree the root, aux, the transcript's data elements, the rate
    elements the indices are read from with their bits and the chains of the two range conditions, per query: j (its bits ARE transcript bits),
    per tree: leaf elements [blocks][16], per level: sib[16], child[16], onehot[16], e_lo[4], e_hi[4]"""

    def __init__(self, layout, statement=None):
        """statement (wrap_arith.Statement + .head: the values the prover absorbs first): build stage B-2 too -- the verifier's field arithmetic
        in two arithmetic templates, and a public input that commits to PUBLIC data only.  None: the hashing-only circuit of rounds 4-5."""
        self.layout, self.statement = layout, statement
        b2 = statement is not None
        c = R1.Circuit(R1.poseidon_template(17))
        self.c = c
        Z = c.new_wire()
        c.add_constraint({Z: 1}, {0: 1}, {})                      # Z = 0
        self.Z = Z
        T = len(layout.trees)
        self.roots = c.new_wires(T)
        self.aux = c.new_wire()                                     # what else the proof is bound to (the aggregator address of the request)
        data = ([] if b2 else list(self.roots)) + [self.aux]       # stage B-2: the roots are private
        # the assignment script (zp_wrap_assign: op, first wire, count, a, b, c): how assign() below fills the caller-set wires
        ops = [(0, 0, 1, 1, 0, 0), (0, Z, 1, 0, 0, 0), (1, self.aux, 1, 0, 0, 0)] + [(2, w, 1, 0, t, 0) for t, w in enumerate(self.roots)]

        # ---- stage B-1: the transcript sponge.  One gadget per absorbed block; the roots in it are the root wires above, the rest new data wires
        # Every gadget's CAPACITY input is a caller-set wire (the prover knows the capacity after every permutation of its own sponge) tied to
        # its predecessor's output by one linear constraint: the gadgets then read caller-set wires only, i.e. they are all evaluated side by
        # side in the first wave of zp_r1cs_eval -- chained through their output wires the 23 + 1 permutations were 24 dependent single-instance
        # launches (16 ms of the 21 ms witness completion at the service's size).
        self.tdata, self.tblocks, self.tcaps = [], [], []          # data wires in absorb order; per block its 16 input wires; the capacity wires
        self.tkind = []                                              # kind of every data wire ("head", "pubs", "evals", "final")
        self.chal = []                                               # stage B-2: per segment but the last, a wire tied to rate element 1 after it
        cap, nblk, prev_out = Z, 0, None

        def next_cap(prev_out, k):
            w = c.new_wire()
            ops.append((15, w, 1, k, 0, 0))
            c.add_constraint({prev_out: 1}, {0: 1}, {w: 1})
            self.tcaps.append(w)
            return w
        for seg in layout.transcript_segments():
            wires, fresh = [], set()
            for kind, cnt in seg:
                if isinstance(kind, tuple):
                    wires.append(self.roots[kind[1]])
                else:
                    new = c.new_wires(cnt)
                    self.tdata += new
                    self.tkind += [kind] * cnt
                    fresh.update(new)
                    wires += new
            for off in range(0, len(wires), 16):
                blk = wires[off:off + 16] + [Z] * (16 - len(wires[off:off + 16]))
                k = 0
                while k < 16:       # the data wires of this block as runs of consecutive wires at consecutive positions (script op 10: block, first position)
                    if blk[k] not in fresh:
                        k += 1
                        continue
                    k2 = k
                    while k2 + 1 < 16 and blk[k2 + 1] in fresh and blk[k2 + 1] == blk[k2] + 1:
                        k2 += 1
                    ops.append((10, blk[k], k2 - k + 1, nblk, k, 0))
                    k = k2 + 1
                if prev_out is not None:
                    cap = next_cap(prev_out, nblk - 1)
                prev_out = c.add_instance([cap] + blk)
                self.tblocks.append(blk)
                nblk += 1
            if b2 and len(self.chal) < len(layout.transcript_segments()) - 1:
                # the challenge squeezed after this segment is read from rate element 1 of the state: a caller-set wire tied to the sponge
                w = c.new_wire()
                ops.append((16, w, 1, len(self.chal), 0, 0))
                c.add_constraint(c.output_lc(prev_out, 1), {0: 1}, {w: 1})
                self.chal.append(w)
        cap = prev_out
        if not b2:
            data += self.tdata
        # the rate the indices are read from: caller-set wires tied to the instance's output state (an R1CS gadget exposes element 0 only)
        self.rates = []
        n_sq = layout.squeeze_perms()
        for k in range(n_sq):
            rate = c.new_wires(16)
            ops.append((11, rate[0], 16, k, 0, 0))
            for i in range(16):
                c.add_constraint(c.output_lc(cap, 1 + i), {0: 1}, {rate[i]: 1})
            self.rates.append(rate)
            if k + 1 < n_sq:
                cap = c.add_instance([next_cap(cap, nblk - 1 + k)] + rate)   # squeeze-only permutation: the whole state goes round
        # canonical bit decompositions of the rate elements that carry an index; 64-bit word w of element e = bits 64 w .. 64 w + 63
        self.ebits = {}
        n_q = layout.n_queries
        for k in range(n_sq):
            for e in range(16):
                if 48 * k + 3 * e >= n_q:
                    break
                bits = c.new_wires(254)
                for b0 in range(0, 254, 64):
                    ops.append((12, bits[b0], min(64, 254 - b0), k, e, b0))
                for b in bits:
                    c.add_constraint({b: 1}, {b: 1}, {b: 1})
                c.add_constraint({b: (1 << i) % R for i, b in enumerate(bits)}, {0: 1}, {self.rates[k][e]: 1})
                # value < r: walking down from the top bit, p = "equal to r so far"; where r has a 0 the value must not have a 1 while equal, where r
                # has a 1 equality continues only through a 1; equal to the end is r itself: excluded
                chain, p = [], 0                                   # p: wire 0 = the constant 1
                for i in range(253, -1, -1):
                    if R_BITS[i]:
                        if p == 0:
                            p = bits[i]
                        else:
                            nxt = c.new_wire()
                            c.add_constraint({p: 1}, {bits[i]: 1}, {nxt: 1})
                            chain.append(nxt)
                            p = nxt
                    else:
                        c.add_constraint({p: 1}, {bits[i]: 1}, {})
                c.add_constraint({p: 1}, {0: 1}, {})
                for o0 in range(0, len(chain), 64):
                    ops.append((13, chain[o0], min(64, len(chain) - o0), k, e, o0))
                assert all(chain[i + 1] == chain[i] + 1 for i in range(len(chain) - 1))
                self.ebits[(k, e)] = bits
                for w in range(3):
                    if 48 * k + 3 * e + w >= n_q:
                        break
                    # the word is used as it stands (no reduction mod p): its top 32 bits are not all set
                    hb = bits[64 * w + 32:64 * w + 64]
                    t = c.new_wires(30)
                    ops.append((14, t[0], 30, k, e, w))
                    c.add_constraint({hb[0]: 1}, {hb[1]: 1}, {t[0]: 1})
                    for i in range(1, 30):
                        c.add_constraint({t[i - 1]: 1}, {hb[i + 1]: 1}, {t[i]: 1})
                    c.add_constraint({t[29]: 1}, {hb[31]: 1}, {})

        self.q = []
        for qi in range(layout.n_queries):
            j = c.new_wire()
            k, e, w = qi // 48, (qi % 48) // 3, qi % 3
            bits = self.ebits[(k, e)][64 * w:64 * w + layout.logm]       # the transcript's bits: nobody chooses an index
            ops.append((3, j, 1, qi, 0, 0))
            c.add_constraint({b: (1 << i) % R for i, b in enumerate(bits)}, {0: 1}, {j: 1})
            if not b2:
                data.append(j)
            trees = []
            for t, (_, width, n_leaves) in enumerate(layout.trees):
                nb = Layout.blocks(width)
                elems = [c.new_wires(16) for _ in range(nb)]
                cap = Z
                for b, blk in enumerate(elems):
                    cap = c.add_instance([cap] + blk)             # leaf sponge: capacity chained
                    if not b2:
                        data += blk
                    ops.append((5, blk[0], 16, qi, t, b))
                cur, lv = cap, []
                a = n_leaves.bit_length() - 1
                for l, _n in enumerate(Layout.levels(n_leaves)):
                    cb = [bits[4 * l + i] if 4 * l + i < a else None for i in range(4)]      # position bits of this level (None: constant 0)
                    sib, child, onehot = c.new_wires(16), c.new_wires(16), c.new_wires(16)
                    e_lo, e_hi = c.new_wires(4), c.new_wires(4)
                    ops += [(6, sib[0], 16, qi, t, l), (6, child[0], 16, qi, t, l), (7, onehot[0], 16, qi, t, l), (8, e_lo[0], 4, qi, t, l), (9, e_hi[0], 4, qi, t, l)]
                    sel = lambda bit, want: ({0: 1} if not want else {}) if bit is None else ({bit: 1} if want else {0: 1, bit: R - 1})
                    for v in range(4):
                        c.add_constraint(sel(cb[0], v & 1), sel(cb[1], v >> 1), {e_lo[v]: 1})
                        c.add_constraint(sel(cb[2], v & 1), sel(cb[3], v >> 1), {e_hi[v]: 1})
                    for kk in range(16):
                        c.add_constraint({e_lo[kk & 3]: 1}, {e_hi[kk >> 2]: 1}, {onehot[kk]: 1})
                        c.add_constraint({onehot[kk]: 1}, {cur: 1, sib[kk]: R - 1}, {child[kk]: 1, sib[kk]: R - 1})     # child = sib + onehot (cur - sib)
                    lv.append({"sib": sib, "child": child, "onehot": onehot, "e_lo": e_lo, "e_hi": e_hi, "bits": cb})
                    cur = c.add_instance([Z] + child)
                c.add_constraint({cur: 1}, {0: 1}, {self.roots[t]: 1})            # the top of the path is the root
                trees.append({"elems": elems, "levels": lv})
            self.q.append({"j": j, "bits": bits, "trees": trees})
        if b2:
            self._stage_b2(c, ops, data)
        # the public input: root of a 16-ary Poseidon tree over the data list (zero padded)
        self.data = data
        level = list(data)
        while len(level) > 1:
            level += [Z] * (-len(level) % 16)
            level = [c.add_instance([Z] + level[i:i + 16]) for i in range(0, len(level), 16)]
        c.add_constraint({level[0]: 1}, {0: 1}, {1: 1}, defines=1)       # the public input IS that root
        self.blob = c.pack()
        hdr = [int.from_bytes(b"PZWRAPS3", "little"), len(ops), sum(o[2] for o in ops), layout.n_queries, T, layout.logm]
        for (_, width, n_leaves) in layout.trees:
            hdr += [width, n_leaves, len(Layout.levels(n_leaves))]
        self.script = np.array(hdr + [len(self.tblocks), n_sq] + [x for o in ops for x in o], dtype=np.uint64)

    def _stage_b2(self, c, ops, data):
        """the verifier's field arithmetic (service/wrap_arith.py) wired to the hashing above, and the list the public input commits to:
        aux | the parameter / statement-digest elements (constants of this circuit when the public inputs enter through their commitment) | the
        commitment to the public inputs | zeta | the statement's sparse fixed columns at zeta.  Everything else -- roots, evaluations, final
        layer, indices, opened values -- is private: the circuit vouches that they exist and verify."""
        from . import wrap_arith as WA
        lay, st = self.layout, self.statement
        sh = WA.Shapes(lay, st)
        assert len(self.chal) == sh.n_chal
        self.zeta_el = c.new_wire()
        ops.append((17, self.zeta_el, 1, 1, 0, 0))                # the low 192 bits of challenge element 1
        self.fz = c.new_wires(sh.n_fz)
        for k, w in enumerate(self.fz):
            ops.append((1, w, 1, 1 + k, 0, 0))                    # aux element 1 + k (the caller computes the fixed columns at zeta natively)
        by_kind = lambda kind: [w for w, k in zip(self.tdata, self.tkind) if k == kind]
        head_w, pubs_w = by_kind("head"), by_kind("pubs")
        from ..stark.prover import PUBLICS_INLINE
        if lay.n_pub > PUBLICS_INLINE:
            hv = [int(v) % WA.P for v in st.head]
            hv += [0] * (-len(hv) % 3)
            elems = [hv[i] + (hv[i + 1] << 64) + (hv[i + 2] << 128) for i in range(0, len(hv), 3)]
            assert len(elems) == len(head_w)
            for w, v in zip(head_w, elems):
                c.add_constraint({w: 1}, {0: 1}, {0: v % R})      # the transcript starts from THIS statement's parameters and digest
        else:
            data += head_w                                         # few public inputs: they sit inside these elements
        data += pubs_w + [self.zeta_el] + self.fz
        g_tpl, exports = WA.global_template(lay, st)
        q_tpl = WA.query_template(lay, st)
        hg, hq = c.add_arith_template(g_tpl), c.add_arith_template(q_tpl)
        g_of = c.add_arith(hg, self.chal + [self.zeta_el] + by_kind("evals") + by_kind("final") + self.fz + (head_w if lay.n_pub <= PUBLICS_INLINE else []))
        exp_w = [g_of(k) for k in exports]
        self.arith_stats = {"global": g_tpl.stats, "query": q_tpl.stats, "global_rows": len(g_tpl.rows), "query_rows": len(q_tpl.rows)}
        for qw in self.q:
            leaf = [w for tw in qw["trees"] for blk in tw["elems"] for w in blk]
            c.add_arith(hq, list(qw["bits"]) + leaf + exp_w)

    # ---- assignment
    def data_values(self, proof, aux, tlog):
        """the list the public input commits to, from a final STARK: roots | aux | transcript data | per query: index, leaf elements of every tree
        (stage B-2 -- aux is then the list [address, fixed columns at zeta ...]: aux | head elements if the publics are inline | publics commitment |
        zeta | fixed columns at zeta)"""
        lay = self.layout
        if self.statement is not None:
            kinds = [k for seg in lay.transcript_segments() for k, cnt in seg for _ in range(cnt) if not isinstance(k, tuple)]
            from ..stark.prover import PUBLICS_INLINE
            out = [int(aux[0]) % R]
            if lay.n_pub <= PUBLICS_INLINE:
                out += [v for v, k in zip(tlog.data, kinds) if k == "head"]
            out += [v for v, k in zip(tlog.data, kinds) if k == "pubs"]
            return out + [tlog.chal[1] & ((1 << 192) - 1)] + [int(v) % R for v in aux[1:]]
        roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
        out = [int(r[0]) for r in roots] + [int(aux) % R] + list(tlog.data)
        if len(proof["queries"]) != lay.n_queries or len(roots) != len(lay.trees):
            raise ValueError("final STARK does not have the shape the wrap circuit was built for")
        for qq in proof["queries"]:
            out.append(int(qq["index"]))
            parts = [qq["trace"], qq["quotient"]] + list(qq["fri"])
            for (name, width, _), part in zip(lay.trees, parts):
                vals = part["values"]
                if len(vals) != width:
                    raise ValueError("opening of %s has the wrong width" % name)
                for b in range(Layout.blocks(width)):
                    out += pack_leaf_block(vals[LEAF_BLOCK * b:LEAF_BLOCK * (b + 1)])
        return out

    def assign(self, proof, aux, tlog):
        """(witness u64[n_wires][4], mask) with every caller-set wire filled from the final STARK and the replay of its transcript (TranscriptLog);
        the gadgets' internal wires, the digests they produce and the public input are left to zp_r1cs_eval -- which refuses (ValueError) when
        the STARK's openings do not hash to its roots or its indices are not the transcript's"""
        from .. import native
        lay = self.layout
        if self.statement is not None:
            vals = {0: 1, self.Z: 0, self.aux: int(aux[0]) % R, self.zeta_el: tlog.chal[1] & ((1 << 192) - 1)}
            assert len(aux) == 1 + len(self.fz)
            for w, v in zip(self.fz, aux[1:]):
                vals[w] = int(v) % R
            for w, v in zip(self.chal, tlog.chal):
                vals[w] = v
        else:
            vals = {0: 1, self.Z: 0, self.aux: int(aux) % R}
        roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
        for w, r in zip(self.roots, roots):
            vals[w] = int(r[0])
        for w, v in zip(self.tdata, tlog.data):
            vals[w] = v
        for k, rate in enumerate(self.rates):
            for w, v in zip(rate, tlog.rates[k]):
                vals[w] = v
        for w, v in zip(self.tcaps, tlog.caps):          # capacity wire k = the capacity after permutation k (the last permutation's is not an input of anything)
            vals[w] = v
        # bits and chains follow the rate elements (wire numbers: bits, then the < r chain, then per used word the 30 partial products)
        for (k, e), bits in self.ebits.items():
            v = tlog.rates[k][e]
            for i, b in enumerate(bits):
                vals[b] = (v >> i) & 1
            w, p = bits[-1] + 1, None
            for i in range(253, -1, -1):
                if R_BITS[i]:
                    if p is None:
                        p = (v >> i) & 1
                    else:
                        p &= (v >> i) & 1
                        vals[w] = p
                        w += 1
            for wd in range(3):
                if 48 * k + 3 * e + wd >= lay.n_queries:
                    break
                acc = (v >> (64 * wd + 32)) & 1
                for i in range(1, 31):
                    acc &= (v >> (64 * wd + 32 + i)) & 1
                    vals[w] = acc
                    w += 1
        for qw, qq in zip(self.q, proof["queries"]):
            j = int(qq["index"])
            vals[qw["j"]] = j
            parts = [qq["trace"], qq["quotient"]] + list(qq["fri"])
            for (name, width, n_leaves), tw, part in zip(lay.trees, qw["trees"], parts):
                for b, blk in enumerate(tw["elems"]):
                    for w, v in zip(blk, pack_leaf_block(part["values"][LEAF_BLOCK * b:LEAF_BLOCK * (b + 1)])):
                        vals[w] = v
                idx = j & (n_leaves - 1)
                path = part["path"]
                if len(path) != len(tw["levels"]):
                    raise ValueError("authentication path of %s has the wrong length" % name)
                for l, (lw, grp) in enumerate(zip(tw["levels"], path)):
                    grp = [int(v) for v in grp]
                    if len(grp) != 16:
                        raise ValueError("malformed authentication path")
                    pos = (idx >> (4 * l)) & 15
                    for k in range(16):
                        vals[lw["sib"][k]] = grp[k]
                        vals[lw["child"][k]] = grp[k]            # child[pos] must equal the digest computed so far: checked by its constraint
                        vals[lw["onehot"][k]] = 1 if k == pos else 0
                    for v in range(4):
                        vals[lw["e_lo"][v]] = 1 if v == (pos & 3) else 0
                        vals[lw["e_hi"][v]] = 1 if v == (pos >> 2) else 0
        n = self.c.n_wires
        w = np.zeros((n, 4), dtype=np.uint64)
        mask = np.zeros(n, dtype=np.uint8)
        ids = np.fromiter(vals.keys(), dtype=np.int64, count=len(vals))
        w[ids] = native.fr_words(list(vals.values()))
        mask[ids] = 1
        return w, mask


def openings_record(proof, layout, tlog):
    """the binary openings record of a BN128-mode proof OBJECT (what zp_stark_openings hands out after zp_stark_prove_bn128; layout at its
    definition in csrc/prove.hip) -- for proofs made by the Python orchestration or the CPU checker, which keep no such record.  tlog: the
    replay of the proof's transcript (TranscriptLog): its absorbed blocks and final rates close the record."""
    w4 = lambda v: [(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
    if len(proof["queries"]) != layout.n_queries or len(roots) != len(layout.trees):
        raise ValueError("final STARK does not have the shape the wrap circuit was built for")
    rec = [int.from_bytes(b"PZOPEN03", "little"), layout.n_queries, len(layout.trees), layout.logm]
    for (_, width, n_leaves) in layout.trees:
        rec += [width, n_leaves, len(Layout.levels(n_leaves))]
    for r in roots:
        rec += w4(r[0])
    for qq in proof["queries"]:
        rec.append(int(qq["index"]))
        for (name, width, n_leaves), part in zip(layout.trees, [qq["trace"], qq["quotient"]] + list(qq["fri"])):
            if len(part["values"]) != width or len(part["path"]) != len(Layout.levels(n_leaves)) or any(len(g) != 16 for g in part["path"]):
                raise ValueError("opening of %s has the wrong shape" % name)
            rec += [int(v) for v in part["values"]]
            for grp in part["path"]:
                for d in grp:
                    rec += w4(d)
    rec += [len(tlog.blocks), len(tlog.rates)]
    for blk in tlog.blocks + tlog.rates:
        for e in blk:
            rec += w4(e)
    assert len(tlog.caps) == len(tlog.blocks) + len(tlog.rates) - 1
    for e in tlog.caps:
        rec += w4(e)
    rec.append(len(tlog.chal))                 # "PZOPEN03": rate element 1 after every segment (the challenges' source; the last one leads the index rates)
    for e in tlog.chal:
        rec += w4(e)
    return np.array(rec, dtype=np.uint64)


def fixed_z_elements(fixed_z):
    """the committed form of the statement's sparse fixed columns at zeta: one field element per column, its three components packed like absorbed
    values (c0 + c1 2^64 + c2 2^128, canonical residues)"""
    return [int(v[0]) % 0xFFFFFFFF00000001 + ((int(v[1]) % 0xFFFFFFFF00000001) << 64) + ((int(v[2]) % 0xFFFFFFFF00000001) << 128) for v in fixed_z]


def zeta_of(chal_element):
    """zeta as the verifier uses it (canonical residues) from challenge element 1: its three low 64-bit words mod p"""
    return [((int(chal_element) >> (64 * k)) & 0xFFFFFFFFFFFFFFFF) % 0xFFFFFFFF00000001 for k in range(3)]


def wrap_circuit(layout, statement=None):
    k = layout.key() + (statement.key() if statement is not None else ())
    if k not in _CACHE:
        _CACHE[k] = WrapCircuit(layout, statement)
    return _CACHE[k]
