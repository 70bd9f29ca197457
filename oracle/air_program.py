"""oracle/air_program.py -- the checker's own reader of a constraint program blob (TEST INFRASTRUCTURE).

The AIR reaches the checker as DATA: the u64 blob specified in include/zeth_prover.h ("constraint program"), the same
bytes zp_eval_quotient interprets on the GPU.  Nothing here imports the product package: the decoder, the row
evaluator and the F_{p^3} evaluator below are written against that specification only, so a bug in the product's
expression lowering, code generator or interpreter cannot hide behind shared code.
PARITY UNPINNED with respect to the external reference prover (see gl_oracle.c)."""
from __future__ import annotations

import hashlib

from . import naive as NV

P = NV.P
MAGIC = int.from_bytes(b"ZPAIR1\0\0", "little")
OP_ADD, OP_SUB, OP_MUL, OP_OUT = 1, 2, 3, 4
K_SLOT, K_COL, K_COLN, K_FIXED, K_PUB, K_CONST, K_XML = range(7)
HEADER_WORDS = 12
STAGE2_WIDTH = {1: 3, 2: 9}     # perm: Z (3 base columns);  lookup: h1, h2, S (9)


class BadProgram(Exception):
    pass


class Program:
    def __init__(self, blob):
        w = [int(v) for v in blob]
        if len(w) < HEADER_WORDS or w[0] != MAGIC:
            raise BadProgram("not a ZPAIR1 constraint program")
        (_, self.width, self.width2, self.n_fixed, self.n_pub, self.n_chal, n_const, n_instr, self.n_constraints,
         self.n_slots, n_s2, self.q_chunks) = w[:HEADER_WORDS]
        if self.n_fixed < 2 or len(w) < HEADER_WORDS + n_const + n_instr + 4 * n_s2:
            raise BadProgram("length does not match the header")
        # fixed columns 2.. : sparse periodic columns, [lp | n_entries << 8] then (pos | is_pub << 63, value or public index) pairs
        self.fixed_cols = []
        fa = HEADER_WORDS + n_const + n_instr + 4 * n_s2
        for _ in range(self.n_fixed - 2):
            if fa >= len(w):
                raise BadProgram("fixed-column table truncated")
            lp, ne = w[fa] & 0xFF, w[fa] >> 8
            if lp > 40 or fa + 1 + 2 * ne > len(w):
                raise BadProgram("fixed-column table truncated")
            ent, seen = [], set()
            for e in range(ne):
                a, v = w[fa + 1 + 2 * e], w[fa + 2 + 2 * e]
                pos, is_pub = a & ((1 << 63) - 1), a >> 63
                if pos >= (1 << lp) or pos in seen or (is_pub and v >= self.n_pub) or (not is_pub and v >= P):
                    raise BadProgram("bad fixed-column entry")
                seen.add(pos)
                ent.append((pos, bool(is_pub), v))
            self.fixed_cols.append((lp, ent))
            fa += 1 + 2 * ne
        if fa != len(w):
            raise BadProgram("length does not match the header")
        at = HEADER_WORDS
        self.consts = w[at:at + n_const]
        at += n_const
        self.instrs = []
        outs = 0
        for x in w[at:at + n_instr]:
            op, dst = x & 0xFF, (x >> 8) & 0xFFFF
            a = ((x >> 24) & 0xF, (x >> 28) & 0xFFFF)
            b = ((x >> 44) & 0xF, (x >> 48) & 0xFFFF)
            if op not in (OP_ADD, OP_SUB, OP_MUL, OP_OUT):
                raise BadProgram("unknown opcode %d" % op)
            for (k, i) in ((a,) if op == OP_OUT else (a, b)):
                lim = {K_SLOT: self.n_slots, K_COL: self.width + self.width2, K_COLN: self.width + self.width2,
                       K_FIXED: self.n_fixed, K_PUB: self.n_pub + self.n_chal, K_CONST: max(n_const, 1), K_XML: 1}.get(k)
                if lim is None or i >= lim:
                    raise BadProgram("operand out of range")
            if op != OP_OUT and dst >= self.n_slots:
                raise BadProgram("destination slot out of range")
            outs += op == OP_OUT
            self.instrs.append((op, dst, a, b))
        if outs != self.n_constraints:
            raise BadProgram("OUT count does not match the header")
        at += n_instr
        self.stage2 = []
        for j in range(n_s2):
            kind, a, b, m = w[at + 4 * j:at + 4 * j + 4]
            if kind not in STAGE2_WIDTH:
                raise BadProgram("unknown stage-2 argument kind")
            self.stage2.append({"kind": kind, "a": a, "b": b, "m": m})
        if sum(STAGE2_WIDTH[s["kind"]] for s in self.stage2) != self.width2 or (self.n_chal not in (0, 3)):
            raise BadProgram("stage-2 widths inconsistent")
        raw = b"".join(int(v).to_bytes(8, "little") for v in w)
        self._sha = hashlib.sha256(raw).digest()

    def digest(self):
        return self._sha.hex()[:16]

    # ---- sparse periodic fixed columns (index k counts from fixed column 2)
    def fixed_period(self, k, pubs):
        """one period of column k as a list of ints"""
        lp, ent = self.fixed_cols[k]
        out = [0] * (1 << lp)
        for pos, is_pub, v in ent:
            out[pos] = int(pubs[v]) % P if is_pub else v
        return out

    def fixed_eval_ext(self, k, pubs, zeta, logn, root32):
        """value at the F_{p^3} point zeta of the degree < N polynomial that agrees with column k on the N = 2^logn trace rows.
        Periodic with period p = 2^lp means f(x) = g(x^(N/p)), g the interpolant of one period over the order-p subgroup:
        g(y) = sum_e v_e (y^p - 1) w_p^pos / (p (y - w_p^pos)) -- a sum over the nonzero entries only."""
        lp, ent = self.fixed_cols[k]
        if lp > logn:
            raise BadProgram("fixed column longer than the trace")
        p = 1 << lp
        y = NV.e3_pow(zeta, 1 << (logn - lp))
        wp = NV.root(lp, root32) if lp else 1
        yp = NV.e3_pow(y, p)
        num = [(yp[0] - 1) % P, yp[1], yp[2]]
        pinv = pow(p, P - 2, P)
        terms = []
        for pos, is_pub, v in ent:
            val = int(pubs[v]) % P if is_pub else v
            if val:
                wj = pow(wp, pos, P)
                terms.append((val * wj % P * pinv % P, [(y[0] - wj) % P, y[1], y[2]]))
        # one inversion for all denominators (Montgomery's trick): prefix products, invert the total, walk back
        pre, run = [], [1, 0, 0]
        for _, den in terms:
            pre.append(run)
            run = NV.e3_mul(run, den)
        inv = NV.e3_inv(run) if terms else [1, 0, 0]
        acc = [0, 0, 0]
        for (c, den), pr in zip(reversed(terms), reversed(pre)):
            dinv = NV.e3_mul(inv, pr)
            inv = NV.e3_mul(inv, den)
            acc = NV.e3_add(acc, [t * c % P for t in dinv])
        return NV.e3_mul(acc, num)

    def digest_words(self):
        return [int.from_bytes(self._sha[8 * i:8 * i + 8], "little") % P for i in range(4)]

    # ---- evaluation with caller-supplied field operations (base field: ints;  F_{p^3}: triples)
    def evaluate(self, col, col_next, fixed, pubs, xml, add, sub, mul, embed):
        """walks the instruction list; returns the constraint values in OUT order"""
        slots = [None] * self.n_slots
        outs = []

        def get(ref):
            k, i = ref
            if k == K_SLOT:
                return slots[i]
            if k == K_COL:
                return col[i]
            if k == K_COLN:
                return col_next[i]
            if k == K_FIXED:
                return fixed[i]
            if k == K_PUB:
                return embed(pubs[i])
            if k == K_CONST:
                return embed(self.consts[i])
            return xml

        for (op, dst, a, b) in self.instrs:
            if op == OP_OUT:
                outs.append(get(a))
            else:
                x, y = get(a), get(b)
                slots[dst] = add(x, y) if op == OP_ADD else sub(x, y) if op == OP_SUB else mul(x, y)
        return outs

    def evaluate_base(self, col, col_next, fixed, pubs, xml):
        return self.evaluate(col, col_next, fixed, pubs, xml, lambda a, b: (a + b) % P, lambda a, b: (a - b) % P,
                             lambda a, b: a * b % P, lambda v: int(v) % P)

    def evaluate_ext(self, col, col_next, fixed, pubs, xml):
        e3_sub = lambda a, b: [(a[i] - b[i]) % P for i in range(3)]
        return self.evaluate(col, col_next, fixed, pubs, xml, NV.e3_add, e3_sub, NV.e3_mul, lambda v: [int(v) % P, 0, 0])
