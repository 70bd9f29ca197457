"""The circuit of the Groth16 wrap (GenFinalProof, proto/prover/v1/prover.proto:130-148; the proof it makes is what
src/settlement/ethereum/mod.rs:338-394 hands to EigenZkVM.verifyBatches): an R1CS over the BN254 scalar field that verifies ALL THE HASHING of
the final STARK's verifier at its queries -- stage A of a recursive-verifier circuit.

Statement (one public input d):  there are, for every query q of the final STARK and every committed tree t (trace, quotient, FRI layers):
the query index j_q with its bits, the leaf's field elements (the opened Goldilocks values packed 56 to a sponge block of 16 elements:
csrc/poseidon_bn254.hip, oracle/naive.py:pack_leaf_block), and per tree level the 16 digests of the group on the path, such that
  * the leaf sponge (width-17 Poseidon-BN254, capacity = element 0) of the leaf elements gives the leaf digest,
  * at every level the digest computed so far sits at position (index >> 4 level) & 15 of the group -- the position selected by the BITS of
    j_q --, and the group hashes ([0, 16 children] -> element 0) to the digest of the next level, the last one being the tree's root,
  * d = the root of a 16-ary Poseidon tree over the list  roots | aux | per query: j_q, then the leaf elements of every tree
    (aux: one more field element the proof is bound to -- the aggregator address of the request).
So d commits to everything the rest of a verifier needs -- roots, indices, opened values -- and the circuit vouches that those values ARE the
leaves under those roots at those indices.  What stays outside (checked natively by whoever holds the final STARK: oracle/wrap_verify.py):
recomputing d from the final STARK's data, the packing of values into elements, the transcript, and the field arithmetic (out-of-domain
identity, DEEP, folds).  Not in the circuit yet: the transcript sponge and the arithmetic -- stage B.

Size at the service's parameters (final STARK: 2^18 rows x 47 columns, blow-up 4, 50 queries, 7 trees): 50 x (7 leaf + 26 node permutations)
+ 354 for the data tree = 2 004 permutation gadgets x 613 constraints + 53 k glue constraints = 1.28 M constraints (2^21 domain), 1.3 M wires."""
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

    def __init__(self, params, W, Wq):
        assert params.hash == "bn128"
        self.params = params
        self.logm = params.logn + params.logb
        M = 1 << self.logm
        gt, qg = bn128_rows_per_leaf_log(W, self.logm), bn128_rows_per_leaf_log(Wq, self.logm)
        self.trees = [("trace", W << gt, M >> gt), ("quotient", Wq << qg, M >> qg)]
        sched, _ = params.fri_schedule()
        for li, (lg, f) in enumerate(sched):
            self.trees.append(("fri%d" % li, 3 << f, 1 << (lg - f)))
        self.n_queries = params.n_queries

    def key(self):
        return (tuple(sorted(self.params.to_dict().items())), tuple(self.trees))

    @staticmethod
    def of_air(air, params):
        """the layout of a BN128-mode proof of `air` (stark/air.py Air) under `params`: the widths come from the STATEMENT, not from a proof text"""
        from ..stark import air as AIR
        assert not air.stage2, "the final STARK's AIR has no stage-2 columns"
        return Layout(params, air.width, 3 * AIR.quotient_chunks(air))

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


class WrapCircuit:
    """the circuit + where its caller-set wires are.  wires: Z (value 0), per tree the root, per query: j, bits, per tree: leaf elements
    [blocks][16], per level: sib[16], child[16], onehot[16], e_lo[4], e_hi[4]"""

    def __init__(self, layout):
        self.layout = layout
        c = R1.Circuit(R1.poseidon_template(17))
        self.c = c
        Z = c.new_wire()
        c.add_constraint({Z: 1}, {0: 1}, {})                      # Z = 0
        self.Z = Z
        T = len(layout.trees)
        self.roots = c.new_wires(T)
        self.aux = c.new_wire()                                     # what else the proof is bound to (the aggregator address of the request)
        data = list(self.roots) + [self.aux]
        # the assignment script (zp_wrap_assign: op, first wire, count, query, tree, block / level): how assign() below fills the caller-set wires
        ops = [(0, 0, 1, 1, 0, 0), (0, Z, 1, 0, 0, 0), (1, self.aux, 1, 0, 0, 0)] + [(2, w, 1, 0, t, 0) for t, w in enumerate(self.roots)]
        self.q = []
        for qi in range(layout.n_queries):
            j = c.new_wire()
            bits = c.new_wires(layout.logm)
            ops += [(3, j, 1, qi, 0, 0), (4, bits[0], len(bits), qi, 0, 0)]
            for b in bits:
                c.add_constraint({b: 1}, {b: 1}, {b: 1})          # bits
            c.add_constraint({b: (1 << k) % R for k, b in enumerate(bits)}, {0: 1}, {j: 1})
            data.append(j)
            trees = []
            for t, (_, width, n_leaves) in enumerate(layout.trees):
                nb = Layout.blocks(width)
                elems = [c.new_wires(16) for _ in range(nb)]
                cap = Z
                for b, blk in enumerate(elems):
                    cap = c.add_instance([cap] + blk)             # leaf sponge: capacity chained
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
                    for k in range(16):
                        c.add_constraint({e_lo[k & 3]: 1}, {e_hi[k >> 2]: 1}, {onehot[k]: 1})
                        c.add_constraint({onehot[k]: 1}, {cur: 1, sib[k]: R - 1}, {child[k]: 1, sib[k]: R - 1})     # child = sib + onehot (cur - sib)
                    lv.append({"sib": sib, "child": child, "onehot": onehot, "e_lo": e_lo, "e_hi": e_hi, "bits": cb})
                    cur = c.add_instance([Z] + child)
                c.add_constraint({cur: 1}, {0: 1}, {self.roots[t]: 1})            # the top of the path is the root
                trees.append({"elems": elems, "levels": lv})
            self.q.append({"j": j, "bits": bits, "trees": trees})
        # the public input: root of a 16-ary Poseidon tree over the data list (zero padded)
        self.data = data
        level = list(data)
        while len(level) > 1:
            level += [Z] * (-len(level) % 16)
            level = [c.add_instance([Z] + level[i:i + 16]) for i in range(0, len(level), 16)]
        c.add_constraint({level[0]: 1}, {0: 1}, {1: 1}, defines=1)       # the public input IS that root
        self.blob = c.pack()
        hdr = [int.from_bytes(b"PZWRAPS1", "little"), len(ops), sum(o[2] for o in ops), layout.n_queries, T, layout.logm]
        for (_, width, n_leaves) in layout.trees:
            hdr += [width, n_leaves, len(Layout.levels(n_leaves))]
        self.script = np.array(hdr + [x for o in ops for x in o], dtype=np.uint64)

    # ---- assignment
    def data_values(self, proof, aux=0):
        """the list the public input commits to, from a final STARK: roots | aux | per query: index, leaf elements of every tree"""
        lay = self.layout
        roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
        out = [int(r[0]) for r in roots] + [int(aux) % R]
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

    def assign(self, proof, aux=0):
        """(witness u64[n_wires][4], mask) with every caller-set wire filled from the final STARK; the gadgets' internal wires, the digests they
        produce and the public input are left to zp_r1cs_eval -- which refuses (ValueError) when the STARK's openings do not hash to its roots"""
        from .. import native
        lay = self.layout
        vals = {0: 1, self.Z: 0, self.aux: int(aux) % R}
        roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
        for w, r in zip(self.roots, roots):
            vals[w] = int(r[0])
        for qw, qq in zip(self.q, proof["queries"]):
            j = int(qq["index"])
            vals[qw["j"]] = j
            for k, b in enumerate(qw["bits"]):
                vals[b] = (j >> k) & 1
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


def openings_record(proof, layout):
    """the binary openings record of a BN128-mode proof OBJECT (what zp_stark_openings hands out after zp_stark_prove_bn128; layout at its
    definition in csrc/prove.hip) -- for proofs made by the Python orchestration or the CPU checker, which keep no such record"""
    w4 = lambda v: [(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    roots = [proof["roots"]["trace"], proof["roots"]["quotient"]] + list(proof["fri"]["roots"])
    if len(proof["queries"]) != layout.n_queries or len(roots) != len(layout.trees):
        raise ValueError("final STARK does not have the shape the wrap circuit was built for")
    rec = [int.from_bytes(b"PZOPEN01", "little"), layout.n_queries, len(layout.trees), layout.logm]
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
    return np.array(rec, dtype=np.uint64)


def wrap_circuit(layout):
    k = layout.key()
    if k not in _CACHE:
        _CACHE[k] = WrapCircuit(layout)
    return _CACHE[k]
