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
        if len(w) != HEADER_WORDS + n_const + n_instr + 4 * n_s2:
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
