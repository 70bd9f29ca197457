"""Goldilocks arithmetic inside an R1CS over the BN254 scalar field -- the gadget layer of wrap stage B-2 (service/wrap_circuit.py).

GenFinalProof's Groth16 proof (proto/prover/v1/prover.proto:130-148; consumed verbatim at src/prover/provider.rs:486-503 and handed to
verifyBatches at src/settlement/ethereum/mod.rs:338-394) has to prove that the final STARK VERIFIES, not only that it hashes: the verifier's
field arithmetic -- out-of-domain identity, DEEP quotient, every fold of every query -- runs over F_p, p = 2^64 - 2^32 + 1, and F_p^3 =
F_p[t] / (t^3 - t - 1), inside a circuit whose native field is F_r (254 bits).  This module builds such circuits as TEMPLATES: a sub-circuit
over local wires (wire 0 = the constant 1, wires 1..n_in inputs, then internal wires), instantiated once per query (50 copies) or once per
proof, exactly like the Poseidon gadget of service/r1cs.py -- so the blob holds every row once, not fifty times.

How a field operation becomes constraints.  A wire holds a NON-NEGATIVE INTEGER below a bound the builder tracks (class L: a linear combination
of wires with integer coefficients and an interval [lo, hi] for its value).  Products and sums are taken over the integers -- r is 254 bits,
a product of two 64-bit values 128: hundreds of products fit in one field element -- and reduced mod p only where the next product would
overflow: `reduce(a)` asks the witness for q, s with a = q p + s, range-checks both by bit decomposition (s < 2^64: a "weak" residue, equal to
a mod p or that plus p; q < 2^k with k from a's bound) and states the identity as ONE linear row.  Equalities are taken mod p
(`assert_zero_mod_p`: a = q p with q range-checked).  Every interval is checked against r when a row is emitted: a row is an identity in F_r
and means the integer identity only while nothing wraps.

The witness is not solved for: every internal wire is DEFINED by an op of the template's witness program (product, quotient and remainder by
p, bits, inverse in F_p^3), run in order by the library (csrc/r1cs.hip: arith_witness) or by `Template.run` below (Python integers: the
reference the tests compare the library with).  An op that cannot be carried out -- a value that does not fit its bits, a zero with no
inverse -- means there is no witness: the statement is false.

Blob section (u64 words; appended to the "PZR1CS02" circuit blob after the explicit constraints, see service/r1cs.py):
  per template: [n_in, n_int, n_coef, n_lc, lc_nnz, n_rows, n_ops, n_inst, first_row, 0, 0, 0]
                coef[n_coef][4] | lc_ptr[n_lc + 1] | lc_ent[lc_nnz] (coefficient id << 32 | local wire) | rows[n_rows][2] (a | b << 32, c)
                | ops[n_ops][4] (opcode | nbits << 8 | flag << 24, first destination wire, a | b << 32, c) | inst[n_inst][n_in + 1] (inputs, base)
  an LC id with bit 31 set is the unit combination of local wire (id & 0x7fffffff); LC 0 of the pool is the empty combination (zero).
"""
from __future__ import annotations

import numpy as np

P = 0xFFFFFFFF00000001
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
UNIT = 1 << 31
OP_MUL, OP_DIVMOD, OP_BITS, OP_INV3 = 1, 3, 4, 5


class NoWitness(ValueError):
    """the witness program cannot be carried out: the statement the template encodes is false for these inputs"""


class L:
    """linear combination over local wires: {wire: integer coefficient} (wire 0 = the constant 1) + the interval its VALUE lies in"""
    __slots__ = ("t", "lo", "hi")

    def __init__(self, t, lo, hi):
        self.t, self.lo, self.hi = t, lo, hi

    def __add__(self, o):
        o = _as_l(o)
        t = dict(self.t)
        for k, v in o.t.items():
            nv = t.get(k, 0) + v
            if nv:
                t[k] = nv
            else:
                t.pop(k, None)
        return L(t, self.lo + o.lo, self.hi + o.hi)

    __radd__ = __add__

    def __neg__(self):
        return L({k: -v for k, v in self.t.items()}, -self.hi, -self.lo)

    def __sub__(self, o):
        return self + (-_as_l(o))

    def __rsub__(self, o):
        return _as_l(o) - self

    def __mul__(self, s):
        s = int(s)
        if s == 0:
            return L({}, 0, 0)
        lo, hi = (self.lo * s, self.hi * s) if s > 0 else (self.hi * s, self.lo * s)
        return L({k: v * s for k, v in self.t.items()}, lo, hi)

    __rmul__ = __mul__

    def within(self, lo, hi):
        """the same combination with a tighter interval the caller can vouch for (e.g. a selection between two values)"""
        assert lo >= self.lo and hi <= self.hi
        return L(self.t, lo, hi)


def _as_l(x):
    if isinstance(x, L):
        return x
    x = int(x)
    return L({0: x} if x else {}, x, x)


class Template:
    """what pack() writes and run() / check() read: plain lists"""

    def __init__(self, n_in, n_wires, coefs, lcs, rows, ops, in_hi):
        self.n_in, self.n_wires, self.coefs, self.lcs, self.rows, self.ops, self.in_hi = n_in, n_wires, coefs, lcs, rows, ops, in_hi
        self.n_int = n_wires - 1 - n_in

    def lc_value(self, lc, w):
        if lc & UNIT:
            return w[lc & (UNIT - 1)]
        return sum(self.coefs[c] * w[k] for c, k in self.lcs[lc]) % R

    def run(self, inputs):
        """the witness program on Python integers: local wire values (list of n_wires ints); NoWitness when an op cannot be carried out"""
        assert len(inputs) == self.n_in
        w = [1] + [int(v) % R for v in inputs] + [None] * self.n_int
        for (op, nbits, flag, dst, a, b, c) in self.ops:
            if op == OP_MUL:
                w[dst] = self.lc_value(a, w) * self.lc_value(b, w) % R
            elif op == OP_DIVMOD:
                v = self.lc_value(a, w)
                q, s = divmod(v, P)
                if flag == 1:                 # weak remainder: the representative in [2^64 - p .. 2^64) is NOT used; s = v mod p
                    pass
                if flag == 2:                 # exact division
                    if s:
                        raise NoWitness("a value that must vanish mod p does not")
                    w[dst] = q
                else:
                    w[dst], w[dst + 1] = q, s
            elif op == OP_BITS:
                v = self.lc_value(a, w)
                if v >> nbits:
                    raise NoWitness("a value does not fit its %d bits" % nbits)
                for i in range(nbits):
                    w[dst + i] = (v >> i) & 1
            elif op == OP_INV3:
                x = [self.lc_value(k, w) % P for k in (a, b, c)]
                inv = e3_inv(x)
                if inv is None:
                    raise NoWitness("zero has no inverse")
                w[dst:dst + 3] = inv
            else:
                raise ValueError("unknown op")
        assert all(v is not None for v in w)
        return w

    def check(self, w):
        """index of the first violated row, or -1"""
        for i, (a, b, c) in enumerate(self.rows):
            if self.lc_value(a, w) * self.lc_value(b, w) % R != self.lc_value(c, w):
                return i
        return -1

    def header(self, n_inst, first_row):
        nnz = sum(len(lc) for lc in self.lcs)
        return [self.n_in, self.n_int, len(self.coefs), len(self.lcs), nnz, len(self.rows), len(self.ops), n_inst, first_row, 0, 0, 0]

    def pack(self, instances, first_row):
        """instances: list of (input global wires, base of the internal wires)"""
        out = self.header(len(instances), first_row)
        for c in self.coefs:
            c %= R
            out += [(c >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
        ptr = [0]
        ent = []
        for lc in self.lcs:
            ent += [(c << 32) | k for c, k in lc]
            ptr.append(len(ent))
        out += ptr + ent
        for (a, b, c) in self.rows:
            out += [a | (b << 32), c]
        for (op, nbits, flag, dst, a, b, c) in self.ops:
            out += [op | (nbits << 8) | (flag << 24), dst, a | (b << 32), c]
        for ins, base in instances:
            assert len(ins) == self.n_in
            out += list(ins) + [base]
        return out


def e3_mul_int(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    d0, d1, d2, d3, d4 = a0 * b0, a0 * b1 + a1 * b0, a0 * b2 + a1 * b1 + a2 * b0, a1 * b2 + a2 * b1, a2 * b2
    return [(d0 + d3) % P, (d1 + d3 + d4) % P, (d2 + d4) % P]          # t^3 = t + 1, t^4 = t^2 + t


def e3_inv(x):
    """inverse in F_p[t] / (t^3 - t - 1) by solving the 3 x 3 system of the multiplication matrix (None for zero)"""
    a0, a1, a2 = [v % P for v in x]
    # columns: x * 1, x * t, x * t^2 expressed in the basis
    m = [[a0, a2, a1], [a1, (a0 + a2) % P, (a1 + a2) % P], [a2, a1, (a0 + a2) % P]]
    rhs = [1, 0, 0]
    n = 3
    m = [row[:] + [rhs[i]] for i, row in enumerate(m)]
    for col in range(n):
        piv = next((r for r in range(col, n) if m[r][col] % P), None)
        if piv is None:
            return None
        m[col], m[piv] = m[piv], m[col]
        iv = pow(m[col][col], P - 2, P)
        m[col] = [v * iv % P for v in m[col]]
        for r in range(n):
            if r != col and m[r][col]:
                f = m[r][col]
                m[r] = [(vr - f * vc) % P for vr, vc in zip(m[r], m[col])]
    return [m[i][3] for i in range(3)]


class Builder:
    """builds one template.  Inputs are declared first (`inp(hi)`: a wire whose value the caller guarantees to be an integer in [0, hi]); every
    other wire is made by a gadget below, which also emits the rows that pin it and the witness op that computes it."""

    def __init__(self):
        self.n_in = 0
        self.n_wires = 1
        self.hi = {0: 1}              # wire -> largest value it can hold (smallest: 0)
        self.rows, self.ops = [], []
        self._sealed = False
        self.stats = {}

    # ---- wires
    def inp(self, hi):
        assert not self._sealed, "inputs are declared before the first gadget"
        self.n_in += 1
        self.n_wires += 1
        self.hi[self.n_wires - 1] = int(hi)
        return self.n_wires - 1

    def _new(self, hi, count=1):
        self._sealed = True
        first = self.n_wires
        self.n_wires += count
        for k in range(count):
            self.hi[first + k] = int(hi)
        return first

    def w(self, wire):
        return L({wire: 1}, 0, self.hi[wire])

    def e3(self, wires):
        return [self.w(k) for k in wires]

    @staticmethod
    def const(v):
        return _as_l(v)

    # ---- rows
    def _row(self, a, b, c, what, mod_r=False):
        """mod_r: the row is MEANT as an identity in F_r (bits_field: the canonical decomposition of a whole field element), no interval check"""
        if not mod_r:
            for x in (a, b, c):
                assert -R < x.lo and x.hi < R, "an interval leaves the field: reduce earlier"
            assert max(abs(a.lo), abs(a.hi)) * max(abs(b.lo), abs(b.hi)) < R, "a product leaves the field: reduce earlier"
        self.rows.append((a, b, c))
        self.stats[what] = self.stats.get(what, 0) + 1

    def _op(self, op, nbits, flag, dst, a=None, b=None, c=None):
        self.ops.append((op, nbits, flag, dst, a, b, c))

    # ---- gadgets
    def mul(self, a, b, what="product"):
        """wire = a b over the integers (both operands non-negative)"""
        a, b = _as_l(a), _as_l(b)
        assert a.lo >= 0 and b.lo >= 0
        w = self._new(a.hi * b.hi)
        self._op(OP_MUL, 0, 0, w, a, b)
        self._row(a, b, self.w(w), what)
        return w

    def assert_bool(self, wire):
        x = self.w(wire)
        self._row(x, x, x, "boolean")
        self.hi[wire] = 1

    def bits(self, a, n, what="bits"):
        """n wires, the bits of a (which must lie in [0, 2^n): the decomposition IS that range check)"""
        a = _as_l(a)
        assert a.lo >= 0
        first = self._new(1, n)
        self._op(OP_BITS, n, 0, first, a)
        acc = L({}, 0, 0)
        for i in range(n):
            x = self.w(first + i)
            self._row(x, x, x, "boolean")
            acc = acc + x * (1 << i)
        self._row(acc - a, _as_l(1), _as_l(0), what)
        return list(range(first, first + n))

    def bits_field(self, wire, what="field element bits"):
        """the CANONICAL 254 bits of a wire that holds an arbitrary field element (a sponge output): sum 2^i b_i = the element in F_r, and the bits
        name a value below r -- walking down from the top bit, p = "equal to r so far": where r has a 0 the value has no 1 while p, where r has
        a 1 p continues through a 1; equal to the end is r itself, excluded.  Without it a value v < 2^254 - r has a second decomposition, v + r."""
        first = self._new(1, 254)
        self._op(OP_BITS, 254, 0, first, self.w(wire))
        acc = L({}, 0, 0)
        for i in range(254):
            x = self.w(first + i)
            self._row(x, x, x, "boolean")
            acc = acc + x * (1 << i)
        self._row(acc - self.w(wire), _as_l(1), _as_l(0), what, mod_r=True)
        p = None
        for i in range(253, -1, -1):
            x = self.w(first + i)
            if (R >> i) & 1:
                p = x if p is None else self.w(self.mul(p, x, what + ": below r"))
            else:
                assert p is not None
                self._row(p, x, _as_l(0), what + ": below r")
        self._row(p, _as_l(1), _as_l(0), what + ": below r")
        return list(range(first, first + 254))

    def reduce(self, a, what="reduction"):
        """a weak residue of a mod p: a wire s in [0, 2^64) with a = q p + s for a range-checked q"""
        a = _as_l(a)
        assert a.lo >= 0, "reduce takes non-negative values: add a multiple of p first (sub_mod)"
        if a.hi < (1 << 64) and len(a.t) == 1 and 0 not in a.t and next(iter(a.t.values())) == 1:
            return next(iter(a.t))              # already a weak wire
        nq = max(1, (a.hi // P).bit_length())
        q = self._new((1 << nq) - 1, 2)
        s = q + 1
        self.hi[s] = (1 << 64) - 1
        self._op(OP_DIVMOD, 0, 0, q, a)
        qb = self.bits(self.w(q), nq, "reduction: quotient bits")
        sb = self.bits(self.w(s), 64, "reduction: remainder bits")
        del qb, sb
        self._row(a - self.w(q) * P - self.w(s), _as_l(1), _as_l(0), what)
        return s

    def assert_zero_mod_p(self, a, what="zero mod p"):
        """a = q p for a range-checked q (a non-negative)"""
        a = _as_l(a)
        assert a.lo >= 0
        nq = max(1, (a.hi // P).bit_length())
        q = self._new((1 << nq) - 1)
        self._op(OP_DIVMOD, 0, 2, q, a)
        self.bits(self.w(q), nq, "zero mod p: quotient bits")
        self._row(a - self.w(q) * P, _as_l(1), _as_l(0), what)

    @staticmethod
    def sub_mod(a, b):
        """a - b plus the multiple of p that keeps it non-negative whatever b is"""
        a, b = _as_l(a), _as_l(b)
        k = -(-max(b.hi, 0) // P)
        return a - b + k * P

    def eq_mod_p(self, a, b, what="equal mod p"):
        self.assert_zero_mod_p(self.sub_mod(a, b), what)

    def select(self, bit, a, b, what="select"):
        """bit ? b : a  as a combination (one product); a, b non-negative"""
        a, b = _as_l(a), _as_l(b)
        assert a.lo >= 0 and b.lo >= 0
        off = a.hi
        t = self.mul(self.w(bit), b - a + off, what)
        out = a + self.w(t) - self.w(bit) * off
        return L(out.t, min(a.lo, b.lo), max(a.hi, b.hi))

    def mux(self, bits, vals, what="mux"):
        """vals[index given by the bit wires, least significant first] as a combination: a tree of selects"""
        vals = [_as_l(v) for v in vals]
        assert len(vals) == 1 << len(bits)
        for b in bits:
            vals = [self.select(b, vals[2 * i], vals[2 * i + 1], what) for i in range(len(vals) // 2)]
        return vals[0]

    # ---- F_p^3 on triples of combinations
    @staticmethod
    def e3_add(a, b):
        return [x + y for x, y in zip(a, b)]

    def e3_sub(self, a, b):
        return [self.sub_mod(x, y) for x, y in zip(a, b)]

    def e3_mul_lazy(self, a, b, what="ext product"):
        """the three unreduced components of a b (nine products; each component a sum of at most five of them)"""
        a, b = [_as_l(x) for x in a], [_as_l(x) for x in b]
        pr = {}
        for i in range(3):
            for j in range(3):
                if a[i].hi == 0 or b[j].hi == 0:
                    pr[(i, j)] = _as_l(0)
                elif not a[i].t.keys() - {0}:            # a constant factor: no product wire
                    pr[(i, j)] = b[j] * a[i].t.get(0, 0)
                elif not b[j].t.keys() - {0}:
                    pr[(i, j)] = a[i] * b[j].t.get(0, 0)
                else:
                    pr[(i, j)] = self.w(self.mul(a[i], b[j], what))
        d = [pr[(0, 0)], pr[(0, 1)] + pr[(1, 0)], pr[(0, 2)] + pr[(1, 1)] + pr[(2, 0)], pr[(1, 2)] + pr[(2, 1)], pr[(2, 2)]]
        return [d[0] + d[3], d[1] + d[3] + d[4], d[2] + d[4]]

    def e3_reduce(self, a, what="ext reduction"):
        return [self.w(self.reduce(x, what)) for x in a]

    def e3_mul(self, a, b, what="ext product"):
        return self.e3_reduce(self.e3_mul_lazy(a, b, what), what + ": reduction")

    def e3_scale_lazy(self, a, s, what="ext scale"):
        """a (ext) times s (base), unreduced"""
        s = _as_l(s)
        if not s.t.keys() - {0}:
            return [x * s.t.get(0, 0) for x in a]
        return [self.w(self.mul(x, s, what)) if x.hi else _as_l(0) for x in a]

    def e3_inv(self, x, what="ext inverse"):
        """y with x y = 1 in F_p^3: three range-checked witness wires, the product checked mod p"""
        x = [_as_l(v) for v in x]
        first = self._new((1 << 64) - 1, 3)
        self._op(OP_INV3, 0, 0, first, x[0], x[1], x[2])
        y = []
        for k in range(3):
            self.bits(self.w(first + k), 64, what + ": range")
            y.append(self.w(first + k))
        pr = self.e3_mul_lazy(x, y, what)
        self.assert_zero_mod_p(self.sub_mod(pr[0], 1), what)
        self.assert_zero_mod_p(pr[1], what)
        self.assert_zero_mod_p(pr[2], what)
        return y

    def e3_eq(self, a, b, what="ext equal"):
        for x, y in zip(a, b):
            self.eq_mod_p(x, y, what)

    def pow_by_bits(self, base_const, bits, start=1, what="power by bits"):
        """start * base_const^(sum bits[i] 2^i) mod p, two bits per step: the factor is a combination of {1, b0, b1, b0 b1} that takes one of
        four constants, the running product is reduced after every step"""
        acc = _as_l(start % P)
        i = 0
        while i < len(bits):
            c1 = pow(base_const, 1 << i, P)
            if i + 1 < len(bits):
                c2 = pow(base_const, 2 << i, P)
                c3 = c1 * c2 % P
                b01 = self.mul(self.w(bits[i]), self.w(bits[i + 1]), what)
                f = _as_l(1) + self.w(bits[i]) * (c1 - 1) + self.w(bits[i + 1]) * (c2 - 1) + self.w(b01) * (c3 - c2 - c1 + 1)
                i += 2
            else:
                f = _as_l(1) + self.w(bits[i]) * (c1 - 1)
                i += 1
            f = L(f.t, 1, P - 1)                         # one of the constants
            if not acc.t.keys() - {0}:
                acc = f * acc.t.get(0, 0)
                acc = self.w(self.reduce(acc, what)) if acc.hi >= (1 << 64) else acc
            else:
                acc = self.w(self.reduce(self.w(self.mul(acc, f, what)), what))
        return acc

    # ---- finish
    def template(self):
        coef_id, coefs = {}, []

        def cid(v):
            v %= R
            if v not in coef_id:
                coef_id[v] = len(coefs)
                coefs.append(v)
            return coef_id[v]
        lc_id, lcs = {(): 0}, [[]]

        def lid(x):
            if x is None:
                return 0
            items = tuple(sorted((k, v % R) for k, v in x.t.items() if v % R))
            if len(items) == 1 and items[0][1] == 1:
                return UNIT | items[0][0]
            if items not in lc_id:
                lc_id[items] = len(lcs)
                lcs.append([(cid(v), k) for k, v in items])
            return lc_id[items]
        rows = [(lid(a), lid(b), lid(c)) for a, b, c in self.rows]
        ops = [(op, nbits, flag, dst, lid(a), lid(b), lid(c)) for (op, nbits, flag, dst, a, b, c) in self.ops]
        in_hi = [self.hi[1 + k] for k in range(self.n_in)]
        return Template(self.n_in, self.n_wires, coefs, lcs, rows, ops, in_hi)
