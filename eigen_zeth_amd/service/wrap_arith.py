"""Wrap stage B-2: the FIELD ARITHMETIC of the final STARK's verifier as two arithmetic templates of the wrap circuit (service/arith.py gadgets;
service/wrap_circuit.py wires them to the hashing of stages A / B-1).

What a verifier of a BN128-mode STARK computes besides hashes (oracle/stark_verify.py states it independently; stark/verifier.py is the product's
native form), and where it now lives:

  GLOBAL template, one instance per proof -- inputs: the rate element behind every challenge (tied to the sponge's state by the caller), the
  transcript's data elements (out-of-domain evaluations, final layer), the committed values of the statement's sparse fixed columns at zeta
    * the challenges alpha, zeta, gamma, beta_l ARE the three low 64-bit words of those rate elements: canonical bit decomposition (< r), the
      words used as they stand (a word >= p is a second name of the same residue: every use below is mod p);
    * unpacking: three 64-bit words per data element, by bits (the decomposition is the range check);
    * the constraint identity at zeta: the statement's constraint program (the u64 blob zp_eval_quotient interprets) instruction by instruction
      over F_p^3, sum_k alpha^k C_k(zeta) = q(zeta) Z_H(zeta), with the two boundary selectors computed here and the SPARSE PERIODIC fixed columns
      taken as committed inputs -- their ~10^5 entries are a function of (statement, public inputs, zeta) alone, so whoever reads the proof's public
      input recomputes them without any part of the STARK (oracle/wrap_verify.py: check_public); evaluating them in the circuit would cost
      more than everything else together;
    * the final FRI layer is low degree (every coefficient of its interpolant from 2^(final_log - logb) on vanishes: linear checks mod p);
    * what every query needs: powers of gamma, the public halves of the two DEEP sums, zeta w.
  QUERY template, one instance per query -- inputs: the index bits (transcript bits), the leaf elements of every tree, the exports above
    * unpacking of the opened values; the row of a grouped leaf and the coset position of a FRI leaf selected by index bits;
    * x = shift w^j from the bits of j; the DEEP quotient  sum_k gamma^k (v_k - ev_k(zeta)) / (x - zeta) + sum_k gamma^(Wall+k) (v_k - ev_k(zeta w)) / (x - zeta w);
    * per FRI layer: the opened coset holds the value the previous layer claims, folds (interpolate 2^f points, evaluate at beta) to the next one's;
      the last fold is the final layer's value at the query's position.
Everything is over the integers inside F_r with reductions mod p where the next product needs them (service/arith.py).  PARITY UNPINNED w.r.t. the
external prover: the protocol is this repo's (stark/prover.py)."""
from __future__ import annotations

from . import arith as AR
from ..stark.prover import bn128_rows_per_leaf_log

P = AR.P
R = AR.R
W64 = (1 << 64) - 1
R_BITS = [(R >> i) & 1 for i in range(254)]

OP_ADD, OP_SUB, OP_MUL, OP_OUT = 1, 2, 3, 4
K_SLOT, K_COL, K_COLN, K_FIXED, K_PUB, K_CONST, K_XML = range(7)
PROGRAM_MAGIC = int.from_bytes(b"ZPAIR1\0\0", "little")


class Statement:
    """what the circuit is built FOR: the constraint program of the final STARK's AIR (decoded from the blob of include/zeth_prover.h) and the
    evaluation domain.  Only the instruction list, the constants and the counts are read: the sparse fixed columns stay outside the circuit."""

    def __init__(self, program, root32, shift, head=None):
        """head: the values the prover absorbs first (service/wrap_circuit.py head_values): the circuit pins the transcript to them"""
        self.head = None if head is None else [int(v) for v in head]
        w = [int(v) for v in program[:12]]
        if w[0] != PROGRAM_MAGIC:
            raise ValueError("not a constraint program")
        (_, self.width, self.width2, self.n_fixed, self.n_pub, self.n_chal, n_const, n_instr, self.n_constraints, self.n_slots, n_s2, self.q_chunks) = w
        if self.width2 or self.n_chal or n_s2:
            raise ValueError("the final STARK's AIR has no stage-2 columns")
        self.consts = [int(v) for v in program[12:12 + n_const]]
        self.instrs = []
        for x in program[12 + n_const:12 + n_const + n_instr]:
            x = int(x)
            self.instrs.append((x & 0xFF, (x >> 8) & 0xFFFF, ((x >> 24) & 0xF, (x >> 28) & 0xFFFF), ((x >> 44) & 0xF, (x >> 48) & 0xFFFF)))
        self.root32, self.shift = int(root32), int(shift)
        import hashlib
        import numpy as np
        self.digest = hashlib.sha256(np.ascontiguousarray(program, dtype=np.uint64).tobytes()).hexdigest()

    def key(self):
        return (self.digest, self.root32, self.shift, tuple(self.head or ()))

    def root(self, logn):
        return pow(self.root32, 1 << (32 - logn), P)


def _wire_of(b, x, what="value"):
    """a wire equal to the combination x (one row)"""
    x = AR._as_l(x)
    if len(x.t) == 1 and 0 not in x.t and next(iter(x.t.values())) == 1:
        return next(iter(x.t))
    return b.mul(x, 1, what)


def _words(b, bits, n):
    """word wires of n consecutive 64-bit groups of bit wires"""
    return [_wire_of(b, sum((b.w(bits[64 * k + i]) * (1 << i) for i in range(min(64, len(bits) - 64 * k))), AR._as_l(0)), "word") for k in range(n)]


def _unpack3(b, el, nwords=3):
    """the 64-bit words of a data element (value < 2^(64 nwords): the decomposition enforces it)"""
    bits = b.bits(b.w(el), 64 * nwords, "unpack")
    return _words(b, bits, nwords)


def _mul_theta(a):
    return [a[2], a[0] + a[2], a[1]]


class Shapes:
    """the counts both templates and the circuit builder agree on, from a wrap_circuit.Layout"""

    def __init__(self, layout, st):
        pr = layout.params
        self.logn, self.logb, self.logm = pr.logn, pr.logb, pr.logn + pr.logb
        self.W, self.Wq = layout.W, layout.Wq
        self.Wall = self.W + self.Wq
        self.sched, self.final_log = pr.fri_schedule()
        self.gt, self.qg = bn128_rows_per_leaf_log(self.W, self.logm), bn128_rows_per_leaf_log(self.Wq, self.logm)
        self.n_chal = 3 + len(self.sched)
        self.n_ev = 2 * self.W + self.Wq
        self.fin_per_plane = -(-(1 << self.final_log) // 3)
        self.n_fz = st.n_fixed - 2
        from ..stark.prover import PUBLICS_INLINE
        self.inline = layout.n_pub <= PUBLICS_INLINE              # few public inputs: absorbed with the parameters, not through a commitment
        self.n_head_el = -(-(layout.n_head + layout.n_pub) // 3) if self.inline else 0
        self.n_pub = layout.n_pub
        self.n_exports = 3 * (self.Wall + self.W) + 3 + 3 + 3 + 3 + 3 * len(self.sched) + 3 * (1 << self.final_log)
        assert st.width == self.W and 3 * st.q_chunks == self.Wq, "statement and layout disagree"


def global_template(layout, st):
    """-> (Template, exports: local wires in the order the query template declares them, zeta word wires)"""
    sh = Shapes(layout, st)
    b = AR.Builder()
    chal_el = [b.inp(R - 1) for _ in range(sh.n_chal)]
    zeta_el = b.inp((1 << 192) - 1)
    ev_el = [b.inp(R - 1) for _ in range(sh.n_ev)]
    fin_el = [b.inp(R - 1) for _ in range(3 * sh.fin_per_plane)]
    fz_el = [b.inp(R - 1) for _ in range(sh.n_fz)]
    head_el = [b.inp(R - 1) for _ in range(sh.n_head_el)]
    pubs = []
    if sh.inline:
        # the public inputs sit behind the parameters in the first absorbed elements: unpack, pin the parameters to this statement's, keep the rest
        words = []
        for e in head_el:
            words += _unpack3(b, e)
        assert st.head is not None and len(st.head) == layout.n_head
        for wv, cv in zip(words, st.head):
            b._row(b.w(wv) - int(cv) % P, AR._as_l(1), AR._as_l(0), "parameters")
        pubs = [b.w(k) for k in words[layout.n_head:layout.n_head + sh.n_pub]]

    # ---- challenges: canonical bits of the rate element, its three low words
    chal, zeta_bits = [], None
    for s, el in enumerate(chal_el):
        bits = b.bits_field(el, "challenge bits")
        chal.append([b.w(k) for k in _words(b, bits[:192], 3)])
        if s == 1:
            zeta_bits = bits[:192]
    alpha, zeta, gamma, betas = chal[0], chal[1], chal[2], chal[3:]
    b._row(sum((b.w(k) * (1 << i) for i, k in enumerate(zeta_bits)), AR._as_l(0)) - b.w(zeta_el), AR._as_l(1), AR._as_l(0), "zeta element")

    # ---- transcript data and committed fixed-column values, unpacked
    ev = [[b.w(k) for k in _unpack3(b, e)] for e in ev_el]
    ev_all, ev_next = ev[:sh.Wall], ev[sh.Wall:]
    n_f = 1 << sh.final_log
    final = []
    for c in range(3):
        words = []
        for e in range(sh.fin_per_plane):
            words += _unpack3(b, fin_el[c * sh.fin_per_plane + e])
        final.append(words[:n_f])
    fz = [[b.w(k) for k in _unpack3(b, e)] for e in fz_el]

    # ---- the constraint identity at zeta
    N = 1 << sh.logn
    wN = st.root(sh.logn)
    wlast = pow(wN, N - 1, P)
    zN = zeta
    for _ in range(sh.logn):
        zN = b.e3_mul(zN, zN, "zeta^N")
    zh = [b.sub_mod(zN[0], 1), zN[1], zN[2]]
    ninv = pow(N, P - 2, P)
    zm1 = [b.sub_mod(zeta[0], 1), zeta[1], zeta[2]]
    xml = [b.sub_mod(zeta[0], wlast), zeta[1], zeta[2]]
    zhn = b.e3_reduce([x * ninv for x in zh], "selectors")
    l_first = b.e3_mul(zhn, b.e3_inv(zm1, "selectors"), "selectors")
    l_last = b.e3_mul(b.e3_reduce([x * wlast for x in zhn], "selectors"), b.e3_inv(xml, "selectors"), "selectors")
    fixed = [l_first, l_last] + fz
    slots = [None] * st.n_slots
    cs = []
    BIG = 1 << 100

    def get(ref):
        k, i = ref
        if k == K_SLOT:
            return slots[i]
        if k == K_COL:
            return ev_all[i]
        if k == K_COLN:
            return ev_next[i]
        if k == K_FIXED:
            return fixed[i]
        if k == K_CONST:
            return [AR._as_l(st.consts[i] % P), AR._as_l(0), AR._as_l(0)]
        if k == K_XML:
            return xml
        if k == K_PUB and sh.inline:
            return [pubs[i], AR._as_l(0), AR._as_l(0)]
        raise ValueError("the statement reads a public input that entered the transcript through a commitment: not available to the wrap circuit")

    def tidy(v):
        return [b.w(b.reduce(x, "program")) if x.hi >= BIG else x for x in v]
    for (op, dst, a, c) in st.instrs:
        if op == OP_OUT:
            cs.append(get(a))
            continue
        x, y = get(a), get(c)
        if op == OP_ADD:
            slots[dst] = tidy(b.e3_add(x, y))
        elif op == OP_SUB:
            slots[dst] = tidy(b.e3_sub(x, y))
        elif op == OP_MUL:
            slots[dst] = b.e3_mul(x, y, "program")
        else:
            raise ValueError("unknown opcode")
    assert len(cs) == st.n_constraints
    lhs = cs[-1]
    for k in range(len(cs) - 2, -1, -1):                  # Horner in alpha
        lhs = b.e3_reduce(b.e3_add(b.e3_mul_lazy(lhs, alpha, "alpha combination"), cs[k]), "alpha combination")
    sNinv = pow(pow(st.shift, N, P), P - 2, P)
    zsN = b.e3_reduce([x * sNinv for x in zN], "quotient at zeta")                    # (zeta / shift)^N
    Q, Wt = st.q_chunks, sh.W
    q = None
    for j in range(Q - 1, -1, -1):
        qj = b.e3_add(b.e3_add(ev_all[Wt + 3 * j], _mul_theta(ev_all[Wt + 3 * j + 1])), _mul_theta(_mul_theta(ev_all[Wt + 3 * j + 2])))
        q = qj if q is None else b.e3_add(b.e3_mul(q, zsN, "quotient at zeta"), qj)
    b.e3_eq(lhs, b.e3_mul_lazy(q, zh, "identity"), "identity at zeta")

    # ---- the final layer is low degree
    w_inv = pow(st.root(sh.final_log), P - 2, P)
    keep = 1 << (sh.final_log - sh.logb)
    for c in range(3):
        for i in range(keep, n_f):
            b.assert_zero_mod_p(sum((b.w(v) * pow(w_inv, (i * k) % n_f, P) for k, v in enumerate(final[c])), AR._as_l(0)), "final layer degree")

    # ---- what the queries share
    n_gp = sh.Wall + sh.W
    gp = [[AR._as_l(1), AR._as_l(0), AR._as_l(0)]]
    for _ in range(n_gp - 1):
        gp.append(b.e3_mul(gp[-1], gamma, "gamma powers"))
    eza = [AR._as_l(0)] * 3
    for k in range(sh.Wall):
        eza = b.e3_add(eza, b.e3_mul_lazy(gp[k], ev_all[k], "DEEP public half"))
    ezb = [AR._as_l(0)] * 3
    for k in range(sh.W):
        ezb = b.e3_add(ezb, b.e3_mul_lazy(gp[sh.Wall + k], ev_next[k], "DEEP public half"))
    eza, ezb = b.e3_reduce(eza, "DEEP public half"), b.e3_reduce(ezb, "DEEP public half")
    zeta_w = b.e3_reduce([x * wN for x in zeta], "zeta w")
    exports = []
    for v in gp + [eza, ezb, zeta, zeta_w] + betas:
        exports += [_wire_of(b, x, "export") for x in v]
    for c in range(3):
        exports += list(final[c])
    assert len(exports) == sh.n_exports
    t = b.template()
    t.stats = dict(b.stats)
    return t, exports


def query_template(layout, st):
    """inputs: logm index bits | per tree the elements of its leaf (16 per sponge block) | the global template's exports"""
    sh = Shapes(layout, st)
    b = AR.Builder()
    jbits = [b.inp(1) for _ in range(sh.logm)]
    leaf_el = []
    for (_, width, _n) in layout.trees:
        nb = max(1, -(-width // 56))
        leaf_el.append([b.inp(R - 1) for _ in range(16 * nb)])
    n_gp = sh.Wall + sh.W
    take3 = lambda: [b.w(b.inp(W64)) for _ in range(3)]
    gp = [take3() for _ in range(n_gp)]
    eza, ezb, zeta, zeta_w = take3(), take3(), take3(), take3()
    betas = [take3() for _ in sh.sched]
    n_f = 1 << sh.final_log
    final = [[b.w(b.inp(W64)) for _ in range(n_f)] for _ in range(3)]

    def leaf_values(elems, width):
        """the `width` opened values of a leaf from its sponge elements (service/wrap_circuit.py pack_leaf_block: element e of a block holds values
        3e, 3e + 1, 3e + 2 and, in bits 192..223, 32-bit half number e of values 48..55)"""
        vals = [None] * width
        for blk in range(max(1, -(-width // 56))):
            base = 56 * blk
            n_here = min(56, width - base)
            halves = {}
            for e in range(16):
                lo = [base + 3 * e + c for c in range(3) if 3 * e + c < min(n_here, 48)]
                hv = 48 + (e >> 1)
                has_half = hv < n_here
                if not lo and not has_half:
                    continue
                nb = 224 if has_half else 64 * len(lo)
                bits = b.bits(b.w(elems[16 * blk + e]), nb, "unpack leaf")
                for c, v in enumerate(lo):
                    vals[v] = b.w(_wire_of(b, sum((b.w(bits[64 * c + i]) * (1 << i) for i in range(64)), AR._as_l(0)), "word"))
                if has_half:
                    halves[e] = sum((b.w(bits[192 + i]) * (1 << i) for i in range(32)), AR._as_l(0))
            for hv in range(48, n_here):
                e0 = 2 * (hv - 48)
                vals[base + hv] = b.w(_wire_of(b, halves[e0] + halves[e0 + 1] * (1 << 32), "word"))
        return vals

    tleaf = leaf_values(leaf_el[0], layout.trees[0][1])
    qleaf = leaf_values(leaf_el[1], layout.trees[1][1])
    # the row of a grouped leaf: column c of row j sits at position (c << g) + (j >> (logm - g))
    tv = [b.mux(jbits[sh.logm - sh.gt:], [tleaf[(c << sh.gt) + s] for s in range(1 << sh.gt)], "row of leaf") if sh.gt else tleaf[c] for c in range(sh.W)]
    qv = [b.mux(jbits[sh.logm - sh.qg:], [qleaf[(c << sh.qg) + s] for s in range(1 << sh.qg)], "row of leaf") if sh.qg else qleaf[c] for c in range(sh.Wq)]
    vals = tv + qv

    # ---- DEEP quotient at x = shift w^j
    wM = st.root(sh.logm)
    x = b.pow_by_bits(wM, jbits, start=st.shift, what="x of the query")
    A = [AR._as_l(0)] * 3
    for k in range(sh.Wall):
        A = b.e3_add(A, b.e3_scale_lazy(gp[k], vals[k], "DEEP sum"))
    B = [AR._as_l(0)] * 3
    for k in range(sh.W):
        B = b.e3_add(B, b.e3_scale_lazy(gp[sh.Wall + k], vals[k], "DEEP sum"))
    A = b.e3_reduce(b.e3_sub(A, eza), "DEEP sum")
    B = b.e3_reduce(b.e3_sub(B, ezb), "DEEP sum")
    d1 = b.e3_inv([b.sub_mod(x, zeta[0]), b.sub_mod(0, zeta[1]), b.sub_mod(0, zeta[2])], "DEEP denominators")
    d2 = b.e3_inv([b.sub_mod(x, zeta_w[0]), b.sub_mod(0, zeta_w[1]), b.sub_mod(0, zeta_w[2])], "DEEP denominators")
    expect = b.e3_reduce(b.e3_add(b.e3_mul_lazy(A, d1, "DEEP quotient"), b.e3_mul_lazy(B, d2, "DEEP quotient")), "DEEP quotient")

    # ---- the FRI layers
    pos = list(jbits)
    cur_shift = st.shift
    for li, (lg, f) in enumerate(sh.sched):
        F = 1 << f
        lv = leaf_values(leaf_el[2 + li], 3 * F)
        pts = [[lv[c * F + k] for c in range(3)] for k in range(F)]
        row_bits, k0_bits = pos[:lg - f], pos[lg - f:lg]
        for c in range(3):
            b.eq_mod_p(b.mux(k0_bits, [pts[k][c] for k in range(F)], "coset position"), expect[c], "layer value")
        w_lg = st.root(lg)
        xinv = b.pow_by_bits(pow(w_lg, P - 2, P), row_bits, start=pow(cur_shift, P - 2, P), what="1 / x of the coset")
        t = b.e3_reduce(b.e3_scale_lazy(betas[li], xinv, "fold"), "fold")
        tp = [[AR._as_l(1), AR._as_l(0), AR._as_l(0)], t]
        for _ in range(2, F):
            tp.append(b.e3_mul(tp[-1], t, "fold"))
        wf_inv = pow(st.root(f), P - 2, P) if f else 1
        finv = pow(F, P - 2, P)
        acc = [AR._as_l(0)] * 3
        for j in range(F):
            cj = [sum((pts[k][c] * (finv * pow(wf_inv, (j * k) % F, P) % P) for k in range(F)), AR._as_l(0)) for c in range(3)]
            acc = b.e3_add(acc, b.e3_mul_lazy(cj, tp[j], "fold"))
        expect = b.e3_reduce(acc, "fold")
        pos = row_bits
        cur_shift = pow(cur_shift, F, P)
    assert len(pos) == sh.final_log
    for c in range(3):
        b.eq_mod_p(b.mux(pos, final[c], "final layer position"), expect[c], "last fold")
    t = b.template()
    t.stats = dict(b.stats)
    return t
