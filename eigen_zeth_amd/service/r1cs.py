"""R1CS over the BN254 scalar field for the Groth16 wrap of GenFinalProof (proto/prover/v1/prover.proto:130-148): circuits made of many copies
of ONE gadget -- a width-t Poseidon-BN254 permutation -- plus explicit glue constraints, in the form the library works on (csrc/r1cs.hip:
zp_r1cs_eval completes and checks a witness and gives A w, B w, C w; zp_r1cs_key_scalars gives the key's scalars).

The gadget (`poseidon_template`): local wire 0 is the constant 1, wires 1..t the input state, then three wires per S-box (x^2, x^4, x^5 of
the S-box input, which is a linear combination of earlier wires plus the round constant: linear layers cost no constraint) and one wire for
element 0 of the output state (the digest of this repo's BN128-hash conventions: a node hashes [0, 16 children], a sponge block [capacity, 16
elements]; both read element 0).  Textbook schedule ARK -> x^5 -> MDS (oracle/naive.py:poseidon_bn254_perm states the same; the tables are
poseidon_constants.bn254_poseidon_params).  t = 17: 8 x 17 + 68 = 204 S-boxes -> 613 constraints, 631 local wires, ~12 000 matrix entries.

Blob layout (u64 words; "PZR1CS01"):
  [0] magic [1] n_wires [2] n_constraints [3] logm [4] t [5] n_local [6] template constraints tc [7] instances [8] extra constraints [9] n_pub, [10..16) 0
  template: def[tc] (the local wire constraint q defines), then for A, B, C: ptr[tc + 1], idx[nnz], val[nnz][4] (standard form)
  instances: per instance t input wires (global), the global index of its first internal wire, its first constraint (= i * tc); then
             waves[n_waves + 1] ([10] = n_waves): the instances of wave k, [waves[k], waves[k + 1]), read only wires that the caller set or an
             earlier wave defined -- they are evaluated in parallel
  extras: def[n_extra] (the global wire extra constraint q defines, ~0: none), then for A, B, C: ptr[n_extra + 1], idx[nnz] (global wires), val[nnz][4]
Constraint numbering: instance i owns [i tc, (i + 1) tc), the extras follow.  Wire 0 = 1, wires 1..n_pub the public inputs.

"PZR1CS02" (round 6, wrap stage B-2): word [11] of the header counts ARITHMETIC TEMPLATES (service/arith.py: Goldilocks arithmetic inside F_r), whose
sections follow the extras in evaluation order; their constraints follow the extras' (template k's instance i owns n_rows consecutive rows from
first_row + i n_rows).  A circuit without such templates is still written as "PZR1CS01"."""
from __future__ import annotations

import numpy as np

from ..poseidon_constants import bn254_poseidon_params

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
MAGIC = int.from_bytes(b"PZR1CS01", "little")
MAGIC2 = int.from_bytes(b"PZR1CS02", "little")
NONE = (1 << 64) - 1
_TEMPLATES = {}


def _lc_add(a, b):
    out = dict(a)
    for k, v in b.items():
        nv = (out.get(k, 0) + v) % R
        if nv:
            out[k] = nv
        else:
            out.pop(k, None)
    return out


def _lc_scale(a, s):
    s %= R
    return {k: v * s % R for k, v in a.items()} if s else {}


class Template:
    """cons: list of (A, B, C, def_wire) with A, B, C dicts {local wire: coefficient}; n_local wires; out = the local wire of output element 0;
    out_lcs[i] = output element i as a linear combination of local wires (the last linear layer over the last round's S-box outputs): a
    circuit that needs more of the output state than element 0 -- a sponge that is SQUEEZED -- ties wires of its own to them
    (Circuit.output_lc)"""

    def __init__(self, t, n_local, cons, out, out_lcs=None):
        self.t, self.n_local, self.cons, self.out = t, n_local, cons, out
        self.n_internal = n_local - 1 - t
        self.out_lcs = out_lcs


def poseidon_template(t=17):
    if t in _TEMPLATES:
        return _TEMPLATES[t]
    rc, mds, rp = bn254_poseidon_params(t)
    state = [{1 + i: 1} for i in range(t)]
    cons, nw = [], 1 + t
    for r in range(8 + rp):
        full = r < 4 or r >= 4 + rp
        after = []
        for i in range(t):
            lc = _lc_add(state[i], {0: rc[r * t + i] % R})
            if full or i == 0:
                x2, x4, x5 = nw, nw + 1, nw + 2
                nw += 3
                cons += [(lc, lc, {x2: 1}, x2), ({x2: 1}, {x2: 1}, {x4: 1}, x4), ({x4: 1}, lc, {x5: 1}, x5)]
                after.append({x5: 1})
            else:
                after.append(lc)
        state = []
        for i in range(t):
            acc = {}
            for j in range(t):
                acc = _lc_add(acc, _lc_scale(after[j], mds[i][j]))
            state.append(acc)
    out = nw
    nw += 1
    cons.append((state[0], {0: 1}, {out: 1}, out))
    _TEMPLATES[t] = Template(t, nw, cons, out, [dict(lc) for lc in state])
    return _TEMPLATES[t]


def _words(v):
    v %= R
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def _csr(rows):
    """rows: list of dicts {wire: coeff} -> (ptr, idx, val words) as flat lists"""
    ptr, idx, val = [0], [], []
    for row in rows:
        for k in sorted(row):
            idx.append(k)
            val += _words(row[k])
        ptr.append(len(idx))
    return ptr + idx + val


class Circuit:
    """instances of one Template wired into global wires + explicit extra constraints.  new_wire() hands out global wire ids; add_instance(inputs)
    returns the global wire of the instance's output; add_constraint(A, B, C) takes dicts over global wires."""

    def __init__(self, template, n_pub=1):
        self.tpl, self.n_pub = template, n_pub
        self.n_wires = 1 + n_pub
        self.instances, self.extras = [], []
        self.ariths = []                 # [(arith.Template, [(input global wires, base of the internal wires)])] in evaluation order
        self._out_wave, self._by_out = {}, {}

    def new_wire(self):
        self.n_wires += 1
        return self.n_wires - 1

    def new_wires(self, n):
        first = self.n_wires
        self.n_wires += n
        return list(range(first, first + n))

    def add_instance(self, inputs):
        """wave of the instance = 1 + the latest wave among the instances whose OUTPUT wires it reads (0 if it reads only caller-set wires)"""
        assert len(inputs) == self.tpl.t and all(0 <= w < self.n_wires for w in inputs)
        base = self.n_wires
        self.n_wires += self.tpl.n_internal
        wave = 1 + max([self._out_wave.get(w, -1) for w in inputs])
        self.instances.append((list(inputs), base, wave))
        out = base + (self.tpl.out - 1 - self.tpl.t)
        self._out_wave[out] = wave
        self._by_out[out] = (list(inputs), base)
        return out

    def output_lc(self, out_wire, i):
        """element i of the output state of the instance whose element-0 wire is `out_wire`, as a linear combination of GLOBAL wires"""
        inputs, base = self._by_out[out_wire]
        return {self.local_to_global(inputs, base, k): v for k, v in self.tpl.out_lcs[i].items()}

    def add_arith_template(self, tpl):
        """registers an arithmetic template (service/arith.py Template); returns its handle for add_arith.  Templates are evaluated in the order
        they are registered: an instance may read internal wires of instances of EARLIER templates (besides caller-set wires)."""
        self.ariths.append((tpl, []))
        return len(self.ariths) - 1

    def add_arith(self, handle, inputs):
        """one instance: `inputs` are global wires (caller-set, or internal wires of an earlier template's instance); returns f(local wire) ->
        global wire for the instance's wires"""
        tpl, insts = self.ariths[handle]
        assert len(inputs) == tpl.n_in and all(0 <= w < self.n_wires for w in inputs)
        base = self.n_wires
        self.n_wires += tpl.n_int
        insts.append((list(inputs), base))
        ins, n_in = list(inputs), tpl.n_in
        return lambda lw: 0 if lw == 0 else ins[lw - 1] if lw <= n_in else base + (lw - 1 - n_in)

    def add_constraint(self, A, B, C, defines=None):
        """defines: a wire (coefficient 1 in C) that this constraint DEFINES when nobody has set it: value = (A w)(B w) - (rest of C w)"""
        assert defines is None or C.get(defines) == 1
        self.extras.append((A, B, C, defines))

    @property
    def n_constraints(self):
        return len(self.instances) * len(self.tpl.cons) + len(self.extras) + sum(len(t.rows) * len(i) for t, i in self.ariths)

    def logm(self):
        lm = 1
        while (1 << lm) < self.n_constraints + 1:      # one row to spare: H has degree m - 2
            lm += 1
        return lm

    def pack(self):
        tpl, tc = self.tpl, len(self.tpl.cons)
        self.instances.sort(key=lambda it: it[2])            # by wave (stable: constraint numbering follows this order)
        n_waves = (self.instances[-1][2] + 1) if self.instances else 1
        bounds = [sum(1 for it in self.instances if it[2] < k) for k in range(n_waves + 1)]
        ariths = [(t, i) for t, i in self.ariths if i]
        hdr = [MAGIC2 if ariths else MAGIC, self.n_wires, self.n_constraints, self.logm(), tpl.t, tpl.n_local, tc, len(self.instances), len(self.extras), self.n_pub,
               n_waves, len(ariths)] + [0] * 4
        body = [c[3] for c in tpl.cons]
        for k in range(3):
            body += _csr([c[k] for c in tpl.cons])
        for i, (inputs, base, _) in enumerate(self.instances):
            body += inputs + [base, i * tc]
        body += bounds
        body += [NONE if e[3] is None else e[3] for e in self.extras]
        for k in range(3):
            body += _csr([e[k] for e in self.extras])
        row = len(self.instances) * tc + len(self.extras)
        for t, insts in ariths:
            body += t.pack(insts, row)
            row += len(t.rows) * len(insts)
        return np.array(hdr + body, dtype=np.uint64)

    # ---- reference evaluation (Python integers): what zp_r1cs_eval does, for tests
    def local_to_global(self, inputs, base, lw):
        return 0 if lw == 0 else inputs[lw - 1] if lw <= self.tpl.t else base + (lw - 1 - self.tpl.t)

    def rows(self):
        """every constraint as (A, B, C) over GLOBAL wires, in blob order"""
        for inputs, base, _ in self.instances:
            for (A, B, C, _) in self.tpl.cons:
                yield tuple({self.local_to_global(inputs, base, k): v for k, v in M.items()} for M in (A, B, C))
        for e in self.extras:
            yield e[:3]
        from .arith import UNIT
        for t, insts in self.ariths:
            for ins, base in insts:
                g = lambda lw: 0 if lw == 0 else ins[lw - 1] if lw <= t.n_in else base + (lw - 1 - t.n_in)
                lc = lambda i: {g(i & (UNIT - 1)): 1} if i & UNIT else {g(k): t.coefs[c] for c, k in t.lcs[i]}
                for (a, b, c) in t.rows:
                    yield lc(a), lc(b), lc(c)

    def complete(self, w):
        """w: dict {global wire: value} of the caller-set wires -> full assignment list, or raises ValueError on a violated constraint"""
        w = dict(w)
        w[0] = 1
        dot = lambda M: sum(c * w[k] for k, c in M.items()) % R
        from .arith import NoWitness
        for t, insts in self.ariths:              # the arithmetic templates' witness programs, in order (they read caller-set wires and earlier templates)
            for ins, base in insts:
                if base in w:
                    continue
                try:
                    loc = t.run([w[k] for k in ins])
                except NoWitness as e:
                    raise ValueError("constraint violated (no witness: %s)" % e)
                if t.check(loc) >= 0:
                    raise ValueError("constraint violated")
                for k in range(t.n_int):
                    w[base + k] = loc[1 + t.n_in + k]
        for inputs, base, _ in sorted(self.instances, key=lambda it: it[2]):
            for (A, B, C, d) in self.tpl.cons:
                g = lambda M: {self.local_to_global(inputs, base, k): v for k, v in M.items()}
                gd = self.local_to_global(inputs, base, d)
                ab = dot(g(A)) * dot(g(B)) % R
                Cg = g(C)
                if gd not in w:
                    w[gd] = (ab - sum(c * w[k] for k, c in Cg.items() if k != gd)) % R
                elif ab != dot(Cg):
                    raise ValueError("constraint violated")
        for (A, B, C, d) in self.extras:
            if d is not None and d not in w:
                w[d] = (dot(A) * dot(B) - sum(c * w[k] for k, c in C.items() if k != d)) % R
            elif dot(A) * dot(B) % R != dot(C):
                raise ValueError("constraint violated")
        return [w[j] for j in range(self.n_wires)]
