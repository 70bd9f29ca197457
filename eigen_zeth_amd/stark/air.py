"""AIR definitions and the constraint code generator.

An AIR is a list of polynomial constraints over the trace columns at the current and the next
row, fixed selector columns, public inputs and the evaluation point x.  Constraints are expression
trees; `emit_quotient_source` turns them into one row-parallel kernel (HIP for gfx950) that evaluates every constraint at an LDE row, combines them with powers of the
F_{p^3} challenge alpha and multiplies by 1/Z_H(x) (periodic with the blow-up).  The reference has
no counterpart (the prover is external to eigen-zeth); the request this serves is
GenChunkProof (src/prover/provider.rs:358-377).

Conventions: transition constraints hold on rows 0..N-2 and are multiplied by (x - w^(N-1));
boundary constraints use the Lagrange selectors L_first / L_last (fixed columns 0 and 1).
Maximum constraint degree 2 in the trace polynomials (+1 for the transition factor), so the
quotient has degree < N and the blow-up can be 2.
"""
from __future__ import annotations

import hashlib

P = 0xFFFFFFFF00000001


class Expr:
    def __add__(self, o): return Op("add", self, wrap(o))
    def __radd__(self, o): return Op("add", wrap(o), self)
    def __sub__(self, o): return Op("sub", self, wrap(o))
    def __rsub__(self, o): return Op("sub", wrap(o), self)
    def __mul__(self, o): return Op("mul", self, wrap(o))
    def __rmul__(self, o): return Op("mul", wrap(o), self)


class Col(Expr):
    def __init__(self, i, nxt=False): self.i, self.nxt = i, nxt
    def key(self): return ("col", self.i, self.nxt)


class Fixed(Expr):
    def __init__(self, i): self.i = i
    def key(self): return ("fixed", self.i)


class Pub(Expr):
    def __init__(self, i): self.i = i
    def key(self): return ("pub", self.i)


class Const(Expr):
    def __init__(self, v): self.v = v % P
    def key(self): return ("const", self.v)


class Chal(Expr):
    """component i of the stage-2 Fiat-Shamir challenge (known only after the first commitment)"""
    def __init__(self, i): self.i = i
    def key(self): return ("chal", self.i)


class XMinusLast(Expr):
    """x - w^(N-1): vanishes on the last trace row only"""
    def key(self): return ("xml",)


class Op(Expr):
    def __init__(self, op, a, b): self.op, self.a, self.b = op, a, b
    def key(self): return (self.op, self.a.key(), self.b.key())


def wrap(v):
    return v if isinstance(v, Expr) else Const(int(v))


L_FIRST, L_LAST = Fixed(0), Fixed(1)


class Air:
    """name, width, number of public inputs, constraint list (transition constraints already carry
    their XMinusLast factor)."""

    def __init__(self, name, width, n_pub, constraints, trace_kind, stage2=None, fixed_cols=None):
        self.name, self.width, self.n_pub = name, width, n_pub
        self.constraints = constraints
        self.trace_kind = trace_kind  # id understood by zp_synth_trace
        # fixed columns 0, 1 are the boundary selectors L_first, L_last; columns 2.. are SPARSE PERIODIC columns known to the
        # verifier: FixedCol(lp, entries) has period 2^lp (lp = logn: not periodic) and is zero except at the listed positions,
        # where it holds a constant or a public input.  Round constants, schedule selectors, per-row expected values (roots,
        # indices) of a verifier AIR (stark/verifier_air.py) are such columns.  The verifier evaluates one at the out-of-domain
        # point by the sparse Lagrange sum over the period-2^lp subgroup; the prover extends one period and tiles it.
        self.fixed_cols = list(fixed_cols) if fixed_cols else []
        self.n_fixed = 2 + len(self.fixed_cols)
        # stage 2 (committed after one F_{p^3} challenge g shared by all arguments): a list of
        #   {"kind": "perm", "a": col, "b": col}             -> grand-product column Z            (3 base columns)
        #   {"kind": "lookup", "a": col, "t": col, "m": col} -> LogUp columns h1, h2, running sum S (9 base columns)
        # laid out one after the other from column index `width`
        self.stage2 = list(stage2) if stage2 else []
        self.width2 = sum(STAGE2_WIDTH[s["kind"]] for s in self.stage2)
        self.n_chal = 3 if stage2 else 0
        self._program = None

    def program(self):
        """the constraint program blob (u64 words, layout in include/zeth_prover.h "constraint program"): the AIR as
        DATA -- what zp_eval_quotient interprets on the GPU and what the checker in oracle/ decodes on its own."""
        if self._program is None:
            self._program = compile_program(self)
        return self._program

    def _sha(self):
        if getattr(self, "_digest", None) is None:       # the program of a verifier AIR is megabytes: hashed once per object
            self._digest = hashlib.sha256(self.program().tobytes()).digest()
        return self._digest

    def digest(self):
        return self._sha().hex()[:16]

    def digest_words(self):
        """the digest as four u64 words (absorbed into the Fiat-Shamir transcript)"""
        h = self._sha()
        return [int.from_bytes(h[8 * i:8 * i + 8], "little") % P for i in range(4)]

    @property
    def symbol(self):
        return "zpair_%s_quotient" % self.name


STAGE2_WIDTH = {"perm": 3, "lookup": 9}


class FixedCol:
    """sparse periodic fixed column: period 2^lp, entries [(pos, value)] with value an int (constant) or Pub(i)"""

    def __init__(self, lp, entries):
        self.lp = int(lp)
        self.entries = sorted(((int(pos), v) for pos, v in entries), key=lambda e: e[0])
        assert all(0 <= pos < (1 << self.lp) for pos, _ in self.entries)
        assert len({pos for pos, _ in self.entries}) == len(self.entries), "duplicate position in a fixed column"

    def values(self, pubs):
        """one period as a list of ints"""
        out = [0] * (1 << self.lp)
        for pos, v in self.entries:
            out[pos] = int(pubs[v.i]) % P if isinstance(v, Pub) else int(v) % P
        return out

    @property
    def public(self):
        return any(isinstance(v, Pub) for _, v in self.entries)


def transition(e):
    return e * XMinusLast()


def fibonacci_air():
    """a' = b, b' = a + b; a[0] = pub0, b[0] = pub1, b[N-1] = pub2"""
    a, b, an, bn = Col(0), Col(1), Col(0, True), Col(1, True)
    cs = [transition(an - b), transition(bn - (a + b)),
          L_FIRST * (a - Pub(0)), L_FIRST * (b - Pub(1)), L_LAST * (b - Pub(2))]
    return Air("fib", 2, 3, cs, trace_kind=0)


def wide_air(width):
    """width columns, degree-2 mixing:  c_i' = c_i * c_(i+1) + c_(i+2) + i   (indices mod width);
    c_i[0] = pub_i for i < 4.  Stands in for the width of a zkEVM chunk trace."""
    cs = []
    for i in range(width):
        c, c1, c2 = Col(i), Col((i + 1) % width), Col((i + 2) % width)
        cs.append(transition(Col(i, True) - (c * c1 + c2 + Const(i))))
    for i in range(min(4, width)):
        cs.append(L_FIRST * (Col(i) - Pub(i)))
    return Air("wide%d" % width, width, min(4, width), cs, trace_kind=1)


def e3x_mul(a, b):
    """product of two F_{p^3} values given as triples of expressions (t^3 = t + 1)"""
    d0 = a[0] * b[0]
    d1 = a[0] * b[1] + a[1] * b[0]
    d2 = a[0] * b[2] + a[1] * b[1] + a[2] * b[0]
    d3 = a[1] * b[2] + a[2] * b[1]
    d4 = a[2] * b[2]
    return [d0 + d3, d1 + d3 + d4, d2 + d4]


def permutation_air():
    """columns a, b, c with  c = a^2  and  b a permutation of a, proven by the grand product
    Z' (b + g) = Z (a + g) (cyclic: the product over all rows is 1), Z[0] = 1.  g = stage-2 challenge."""
    a, b, c = Col(0), Col(1), Col(2)
    Z = [Col(3), Col(4), Col(5)]
    Zn = [Col(3, True), Col(4, True), Col(5, True)]
    g = [Chal(0), Chal(1), Chal(2)]
    lhs = e3x_mul(Zn, [b + g[0], g[1], g[2]])
    rhs = e3x_mul(Z, [a + g[0], g[1], g[2]])
    cs = [c - a * a] + [lhs[i] - rhs[i] for i in range(3)]
    cs += [L_FIRST * (Z[0] - 1), L_FIRST * Z[1], L_FIRST * Z[2], L_FIRST * (a - Pub(0))]
    return Air("perm", 3, 1, cs, trace_kind=2, stage2=[{"kind": "perm", "a": 0, "b": 1}])


def chunk_air(width):
    """The synthetic chunk AIR of SURVEY.md 8d (C3): wide mixing columns + Fibonacci + permutation + range check.
    columns 0..Ww-1 (Ww = width-8): the wide_air mix;  then  fa, fb (Fibonacci),  r (values < 2^k),  q (a permutation
    of r, grand product),  t (range table: starts at 0, steps by 0 or 1, ends at pub7 = 2^k-1),  m (multiplicities of
    the LogUp lookup r in t),  c = r^2,  d = fa*r + fb.   Stage 2 (challenge g): Z | h1, h2, S  (12 base columns)."""
    assert width >= 12
    Ww = width - 8
    cs = []
    for i in range(Ww):
        c, c1, c2 = Col(i), Col((i + 1) % Ww), Col((i + 2) % Ww)
        cs.append(transition(Col(i, True) - (c * c1 + c2 + Const(i))))
    for i in range(4):
        cs.append(L_FIRST * (Col(i) - Pub(i)))
    fa, fb, r, q, t, m, c, d = [Col(Ww + i) for i in range(8)]
    fan, fbn, tn = Col(Ww, True), Col(Ww + 1, True), Col(Ww + 4, True)
    cs += [transition(fan - fb), transition(fbn - (fa + fb)),
           L_FIRST * (fa - Pub(4)), L_FIRST * (fb - Pub(5)), L_LAST * (fb - Pub(6))]
    cs += [c - r * r, d - (fa * r + fb)]
    step = tn - t
    cs += [L_FIRST * t, L_LAST * (t - Pub(7)), transition(step * (step - 1))]
    g = [Chal(0), Chal(1), Chal(2)]
    s2 = lambda j, nxt=False: [Col(width + 3 * j + i, nxt) for i in range(3)]
    Z, Zn, H1, H2, S, Sn = s2(0), s2(0, True), s2(1), s2(2), s2(3), s2(3, True)
    lhs = e3x_mul(Zn, [q + g[0], g[1], g[2]])
    rhs = e3x_mul(Z, [r + g[0], g[1], g[2]])
    cs += [lhs[i] - rhs[i] for i in range(3)]
    cs += [L_FIRST * (Z[0] - 1), L_FIRST * Z[1], L_FIRST * Z[2]]
    p1 = e3x_mul(H1, [r + g[0], g[1], g[2]])
    p2 = e3x_mul(H2, [t + g[0], g[1], g[2]])
    cs += [p1[0] - 1, p1[1], p1[2], p2[0] - m, p2[1], p2[2]]
    cs += [Sn[i] - S[i] - H1[i] + H2[i] for i in range(3)]      # cyclic: the sum over all rows is zero
    cs += [L_FIRST * S[i] for i in range(3)]
    return Air("chunk%d" % width, width, 8, cs, trace_kind=3,
               stage2=[{"kind": "perm", "a": Ww + 2, "b": Ww + 3}, {"kind": "lookup", "a": Ww + 2, "t": Ww + 4, "m": Ww + 5}])


def cubic_air():
    """degree-3 transition constraints (quotient committed in two pieces at blow-up 2):
    a' = a^3 + b,  b' = a b + 7;  a[0] = pub0, b[0] = pub1, a[N-1] = pub2"""
    a, b, an, bn = Col(0), Col(1), Col(0, True), Col(1, True)
    cs = [transition(an - (a * a * a + b)), transition(bn - (a * b + Const(7))),
          L_FIRST * (a - Pub(0)), L_FIRST * (b - Pub(1)), L_LAST * (a - Pub(2))]
    return Air("cubic", 2, 3, cs, trace_kind=None)


def cubic_witness(logn, seed):
    """(trace u64[2][N], publics) of cubic_air: host-side stand-in like zp_synth_trace"""
    import numpy as np
    N = 1 << logn
    tr = np.zeros((2, N), dtype=np.uint64)
    a, b = (seed * 0x9E3779B97F4A7C15 + 1) % P, (seed * 0xC2B2AE3D27D4EB4F + 5) % P
    a0, b0 = a, b
    for i in range(N):
        tr[0, i], tr[1, i] = a, b
        a, b = (a * a * a + b) % P, (a * b + 7) % P
    return tr, np.array([a0, b0, int(tr[0, N - 1])], dtype=np.uint64)


def periodic_air(logn):
    """exercises the sparse periodic fixed columns: K (period 4, dense constants), S (period 8, one nonzero entry), PV (not
    periodic: a public input at one row).   a' = a K + b,  b' = b^2 + K  (transitions);  a[0] = pub0;
    S (a - b) = S c  (c carries a - b at every 8th row, anything elsewhere);  PV = L_first pub1  (identity between a
    public-entry column and a selector: the prover's extension and the verifier's sparse evaluation must agree)."""
    a, b, c, an, bn = Col(0), Col(1), Col(2), Col(0, True), Col(1, True)
    K, S, PV = Fixed(2), Fixed(3), Fixed(4)
    cs = [transition(an - (a * K + b)), transition(bn - (b * b + K)), L_FIRST * (a - Pub(0)), S * (a - b) - S * c,
          PV - L_FIRST * Pub(1)]
    fc = [FixedCol(2, [(0, 3), (1, 5), (2, 7), (3, 11)]), FixedCol(3, [(7, 1)]), FixedCol(logn, [(0, Pub(1))])]
    return Air("periodic%d" % logn, 3, 2, cs, trace_kind=None, fixed_cols=fc)


def periodic_witness(logn, seed):
    import numpy as np
    N = 1 << logn
    tr = np.zeros((3, N), dtype=np.uint64)
    a, b = (seed * 0x9E3779B97F4A7C15 + 1) % P, (seed * 0xC2B2AE3D27D4EB4F + 5) % P
    a0 = a
    kk = [3, 5, 7, 11]
    for i in range(N):
        tr[0, i], tr[1, i] = a, b
        tr[2, i] = (a - b) % P if i % 8 == 7 else (i * 977 + seed) % P
        a, b = (a * kk[i % 4] + b) % P, (b * b + kk[i % 4]) % P
    return tr, np.array([a0, (seed * 31 + 9) % P], dtype=np.uint64)


BUILTIN_AIRS = {"cubic": cubic_air, "perm": permutation_air, "fib": fibonacci_air, "wide8": lambda: wide_air(8), "wide32": lambda: wide_air(32),
                "wide64": lambda: wide_air(64), "chunk16": lambda: chunk_air(16), "chunk64": lambda: chunk_air(64)}


_BUILTIN_CACHE = {}


def get_air(name):
    """the built-in AIR of that name -- ONE object per name (its program blob and digest are computed once; callers treat AIRs as read-only)"""
    if name not in _BUILTIN_CACHE:
        _BUILTIN_CACHE[name] = BUILTIN_AIRS[name]()
    return _BUILTIN_CACHE[name]


# ---------------------------------------------------------------------------------- constraint program (AIR as data)
PROGRAM_MAGIC = int.from_bytes(b"ZPAIR1\0\0", "little")
OP_ADD, OP_SUB, OP_MUL, OP_OUT = 1, 2, 3, 4
K_SLOT, K_COL, K_COLN, K_FIXED, K_PUB, K_CONST, K_XML = 0, 1, 2, 3, 4, 5, 6
S2_PERM, S2_LOOKUP = 1, 2
PROGRAM_HEADER_WORDS = 12


def degree(e):
    """(a, b): the polynomial e has degree <= a*(N-1) + b when every column is a polynomial of degree N-1"""
    if isinstance(e, (Col, Fixed)):
        return (1, 0)
    if isinstance(e, XMinusLast):
        return (0, 1)
    if isinstance(e, (Pub, Const, Chal)):
        return (0, 0)
    da, db = degree(e.a), degree(e.b)
    if e.op == "mul":
        return (da[0] + db[0], da[1] + db[1])
    return max(da, db)


def quotient_chunks(air):
    """number of degree-<N pieces the quotient splits into: deg(C/Z_H) = (a-1)N - a + b"""
    q = 1
    for c in air.constraints:
        a, b = degree(c)
        q = max(q, a - 1 if b < a else a)
    return q


def compile_program(air):
    """Expr trees -> common-subexpression-eliminated three-address code over a small slot file -> u64 blob.
    Constraint k is announced by an OUT instruction right after the value it names exists, so an interpreter can
    fold it into its alpha-accumulators at once and no constraint value has to stay live."""
    import numpy as np
    consts, cmap = [], {}
    instrs, memo = [], {}          # instrs: [op, a_ref, b_ref]; the value of instruction k is SSA name k

    def ref(e):
        k = e.key()
        if k in memo:
            return memo[k]
        if isinstance(e, Col):
            r = (K_COLN if e.nxt else K_COL, e.i)
        elif isinstance(e, Fixed):
            r = (K_FIXED, e.i)
        elif isinstance(e, Pub):
            r = (K_PUB, e.i)
        elif isinstance(e, Chal):
            r = (K_PUB, air.n_pub + e.i)          # challenges follow the publics
        elif isinstance(e, Const):
            if e.v not in cmap:
                cmap[e.v] = len(consts)
                consts.append(e.v)
            r = (K_CONST, cmap[e.v])
        elif isinstance(e, XMinusLast):
            r = (K_XML, 0)
        else:
            a, b = ref(e.a), ref(e.b)
            instrs.append([{"add": OP_ADD, "sub": OP_SUB, "mul": OP_MUL}[e.op], a, b])
            r = ("ssa", len(instrs) - 1)
        memo[k] = r
        return r

    for c in air.constraints:
        r = ref(c)
        instrs.append([OP_OUT, r, (K_CONST, 0)])
    # slot allocation: SSA value k lives from instruction k to its last use
    last = {}
    for k, (_, a, b) in enumerate(instrs):
        for r in (a, b):
            if r[0] == "ssa":
                last[r[1]] = k
    expiring = {}
    for v, k in last.items():
        expiring.setdefault(k, []).append(v)
    free, slot_of, n_slots, words = [], {}, 0, []
    for k, (op, a, b) in enumerate(instrs):
        (ka, ia), (kb, ib) = [((K_SLOT, slot_of[r[1]]) if r[0] == "ssa" else r) for r in (a, b)]
        for v in expiring.get(k, []):     # operands that die here free their slot first: dst may reuse a source slot
            free.append(slot_of[v])
        d = 0
        if op != OP_OUT:
            assert k in last, "dead value after CSE"
            if free:
                d = free.pop()
            else:
                d = n_slots
                n_slots += 1
            slot_of[k] = d
        assert max(ia, ib, d) < (1 << 16) and max(ka, kb) < 16
        words.append(op | (d << 8) | (ka << 24) | (ia << 28) | (kb << 44) | (ib << 48))
    s2 = []
    for st in air.stage2:
        if st["kind"] == "perm":
            s2 += [S2_PERM, st["a"], st["b"], 0]
        else:
            s2 += [S2_LOOKUP, st["a"], st["t"], st["m"]]
    fx = []                       # sparse periodic fixed columns: [lp | n_entries << 8], then (pos | is_pub << 63, value or public index) pairs
    for fc in air.fixed_cols:
        assert fc.lp < 64 and len(fc.entries) < (1 << 40)
        fx.append(fc.lp | (len(fc.entries) << 8))
        for pos, v in fc.entries:
            if isinstance(v, Pub):
                assert v.i < air.n_pub
                fx += [pos | (1 << 63), v.i]
            else:
                fx += [pos, int(v) % P]
    hdr = [PROGRAM_MAGIC, air.width, air.width2, air.n_fixed, air.n_pub, air.n_chal, len(consts), len(words),
           len(air.constraints), max(n_slots, 1), len(air.stage2), quotient_chunks(air)]
    assert len(hdr) == PROGRAM_HEADER_WORDS
    return np.array(hdr + consts + words + s2 + fx, dtype=np.uint64)


# ---------------------------------------------------------------------------------- code generation
class _Emitter:
    """common-subexpression-eliminating emitter of straight-line code over u64 field elements"""

    def __init__(self, n_pub=0):
        self.n_pub = n_pub
        self.lines, self.memo, self.n = [], {}, 0
        self.cols, self.fixed = set(), set()

    def tmp(self, expr_c):
        name = "t%d" % self.n
        self.n += 1
        self.lines.append("    const u64 %s = %s;" % (name, expr_c))
        return name

    def emit(self, e):
        k = e.key()
        if k in self.memo:
            return self.memo[k]
        if isinstance(e, Col):
            self.cols.add((e.i, e.nxt))
            r = "c%d%s" % (e.i, "n" if e.nxt else "")
        elif isinstance(e, Fixed):
            self.fixed.add(e.i)
            r = "f%d" % e.i
        elif isinstance(e, Pub):
            r = "pub[%d]" % e.i
        elif isinstance(e, Const):
            r = "%dULL" % e.v
        elif isinstance(e, Chal):
            r = "pub[%d]" % (self.n_pub + e.i)   # challenges follow the publics in the same array
        elif isinstance(e, XMinusLast):
            r = "xml"
        else:
            a, b = self.emit(e.a), self.emit(e.b)
            r = self.tmp("gl_%s(%s, %s)" % (e.op, a, b))
        self.memo[k] = r
        return r


def emit_quotient_source(air, target="hip"):
    """device kernel + host launchers exported as air.symbol (the whole evaluation domain) and air.symbol + "_rows" (a row window of a sharded
    proof: explicit strides, the window's first row, halo rows behind every column unless the window is the whole domain) -- ONE kernel serves both.
    gfx950 only; the checker in oracle/ interprets the constraint program blob instead and shares no code with this emitter."""
    assert target == "hip"
    em = _Emitter(air.n_pub)
    outs = [em.emit(c) for c in air.constraints]
    body = []
    for (i, nxt) in sorted(em.cols):
        body.append("    const u64 c%d%s = cols[(u64)%d * sc + %s];" % (i, "n" if nxt else "", i, "in" if nxt else "i"))
    # fixed columns 0 / 1 (the boundary selectors) are whole columns (a window of them for a row shard); the sparse periodic ones behind them are ONE
    # extended period each (zp_fixed_columns: column i >= 2 of period 2^lp holds 2^lp * b values at offset 2 * stride + b * sum of the earlier
    # periods; domain row r reads r mod 2^lp b)
    prefix, acc = {}, 0
    for k, fc in enumerate(air.fixed_cols):
        prefix[2 + k] = (acc, 1 << fc.lp)
        acc += 1 << fc.lp
    for i in sorted(em.fixed):
        if i < 2:
            body.append("    const u64 f%d = fixedc[(u64)%d * sf + i];" % (i, i))
        else:
            off, per = prefix[i]
            body.append("    const u64 f%d = fixedc[2 * sf + b * %dULL + (r & (b * %dULL - 1))];" % (i, off, per))
    body += em.lines
    # random linear combination with alpha^k as three unreduced 160-bit dot products (gl_acc), reduced once
    body.append("    gl_acc s0 = gl_acc_zero(), s1 = gl_acc_zero(), s2 = gl_acc_zero();")
    for k, o in enumerate(outs):
        body.append("    gl_acc_mac(s0, %s, apow[%d]); gl_acc_mac(s1, %s, apow[%d]); gl_acc_mac(s2, %s, apow[%d]);"
                    % (o, 3 * k, o, 3 * k + 1, o, 3 * k + 2))
    body.append("    const u64 a0 = gl_acc_reduce(s0), a1 = gl_acc_reduce(s1), a2 = gl_acc_reduce(s2);")
    body.append("    const u64 zi = zhinv[r & (b - 1)];")
    body.append("    out[i] = gl_mul(a0, zi); out[so + i] = gl_mul(a1, zi); out[2 * so + i] = gl_mul(a2, zi);")
    body = "\n".join(body)
    hdr = ("// generated by eigen_zeth_amd/stark/air.py for AIR '%s' (digest %s) -- do not edit\n"
           "// row-parallel constraint evaluation + quotient: lane = LDE row, column reads coalesced.\n"
           % (air.name, air.digest()))
    tail = ("const u64 *__restrict__ pub, const u64 *__restrict__ apow, const u64 *__restrict__ zhinv, const u64 *__restrict__ xs_lo, "
            "const u64 *__restrict__ xs_hi, int lb, u64 shift, u64 wlast, u64 *__restrict__ out")
    args = "const u64 *__restrict__ cols, const u64 *__restrict__ fixedc, u64 M, u64 b, " + tail
    # the window form: rows [row0, row0 + nrows) of the M-row domain; columns at stride sc (with b halo rows behind each unless the window is the
    # whole domain: then the next row wraps), the two selectors at stride sf, the output planes at stride so
    kargs = "const u64 *__restrict__ cols, u64 sc, const u64 *__restrict__ fixedc, u64 sf, u64 M, u64 b, u64 row0, u64 nrows, " + tail + ", u64 so"
    xcode = ("    const u64 r = row0 + i;\n"
             "    const u64 in = nrows == M ? ((i + b) & (M - 1)) : i + b;\n"
             "    const u64 x = gl_mul(shift, gl_mul(xs_lo[r & ((1ULL << lb) - 1)], xs_hi[r >> lb]));\n"
             "    const u64 xml = gl_sub(x, wlast);\n")
    launch = ("    hipLaunchKernelGGL(%s_kernel, dim3((unsigned)((%s + 255) / 256)), dim3(256), 0, (hipStream_t)stream,\n"
              "                       cols, %s, fixedc, %s, M, b, %s, %s, pub, apow, zhinv, xs_lo, xs_hi, lb, shift, wlast, out, %s);\n"
              "    return (int)hipGetLastError();\n}\n")
    return (hdr + '#include <hip/hip_runtime.h>\n#include "gl.hpp"\n'
            "__global__ void __launch_bounds__(256) %s_kernel(%s) {\n"
            "    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;\n    if (i >= nrows) return;\n%s%s\n}\n"
            % (air.symbol, kargs, xcode, body)
            + 'extern "C" int %s(void *stream, %s) {\n' % (air.symbol, args)
            + launch % (air.symbol, "M", "M", "M", "(u64)0", "M", "M")
            + 'extern "C" int %s_rows(void *stream, const u64 *cols, u64 sc, const u64 *fixedc, u64 sf, u64 M, u64 b, u64 row0, u64 nrows, %s, u64 so) {\n'
            % (air.symbol, tail.replace("__restrict__ ", ""))
            + launch % (air.symbol, "nrows", "sc", "sf", "row0", "nrows", "so"))
