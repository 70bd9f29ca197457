"""Goldilocks arithmetic inside F_r (service/arith.py) and arithmetic templates in the circuit blob (service/r1cs.py "PZR1CS02", csrc/r1cs.hip) -- the
machinery of wrap stage B-2, on the CPU: the gadgets compute what integer arithmetic mod p computes; a tampered wire violates a row; an impossible
op (the inverse of zero, a value that does not fit its bits) is "no witness"; the library's host evaluator (witness programs + rows) and its key
scalars equal the Python reference / the definition."""
import random

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.service import arith as AR
from eigen_zeth_amd.service import r1cs as R1

P, R = AR.P, AR.R
W64 = (1 << 64) - 1


def _template():
    b = AR.Builder()
    a = [b.inp(W64) for _ in range(3)]
    c = [b.inp(W64) for _ in range(3)]
    bits = [b.inp(1) for _ in range(5)]
    vals = [b.inp(W64) for _ in range(8)]
    el = b.inp(R - 1)
    A, C = b.e3(a), b.e3(c)
    out = {"mul": b.e3_mul(A, C), "inv": b.e3_inv(A), "sub": b.e3_reduce(b.e3_sub(A, C)), "pow": b.pow_by_bits(7, bits, start=49),
           "mux": b.mux(bits[:3], [b.w(v) for v in vals]), "elbits": b.bits_field(el)}
    b.e3_eq(b.e3_mul(out["inv"], A), [1, 0, 0])
    return b.template(), out


def test_gadgets_compute_field_arithmetic_and_pin_every_wire():
    T, out = _template()
    rnd = random.Random(5)
    for trial in range(4):
        av = [rnd.randrange(P) for _ in range(3)]
        cv = [rnd.randrange(1 << 64) for _ in range(3)]        # weak residues: any 64-bit value
        bv = [rnd.randrange(2) for _ in range(5)]
        vv = [rnd.randrange(1 << 64) for _ in range(8)]
        ev = rnd.randrange(R) if trial else R - 1
        w = T.run(av + cv + bv + vv + [ev])
        assert T.check(w) == -1
        g = lambda x: sum(co * w[k] for k, co in x.t.items())
        assert [g(x) % P for x in out["mul"]] == AR.e3_mul_int(av, cv)
        assert [g(x) % P for x in out["inv"]] == AR.e3_inv(av) and AR.e3_mul_int(AR.e3_inv(av), av) == [1, 0, 0]
        assert [g(x) % P for x in out["sub"]] == [(x - y) % P for x, y in zip(av, cv)]
        assert g(out["pow"]) % P == 49 * pow(7, sum(bv[i] << i for i in range(5)), P) % P
        assert g(out["mux"]) == vv[bv[0] + 2 * bv[1] + 4 * bv[2]]
        assert sum(w[k] << i for i, k in enumerate(out["elbits"])) == ev
        for _ in range(40):                                     # every internal wire is pinned: changing one violates a row
            k = rnd.randrange(T.n_in + 1, T.n_wires)
            w2 = list(w)
            w2[k] = (w2[k] + 1) % R
            assert T.check(w2) >= 0
        # the second decomposition of a small field element (value + r < 2^254) satisfies the recomposition row mod r but not the "< r" chain
        if ev + R < (1 << 254):
            w3 = list(w)
            for i, k in enumerate(out["elbits"]):
                w3[k] = ((ev + R) >> i) & 1
            assert T.check(w3) >= 0
    with pytest.raises(AR.NoWitness, match="inverse"):
        T.run([0, 0, 0] + cv + bv + vv + [ev])


def test_builder_refuses_arithmetic_that_would_wrap_the_field():
    b = AR.Builder()
    x = b.inp(R - 1)
    with pytest.raises(AssertionError, match="leaves the field"):
        b.mul(b.w(x), b.w(x))
    with pytest.raises(AssertionError):
        b.bits(b.w(x), 254)                                     # a whole field element needs bits_field (canonical decomposition)


def test_templates_in_the_circuit_blob_host_evaluator_and_key_scalars():
    T, out = _template()
    rnd = random.Random(9)
    c = R1.Circuit(R1.poseidon_template(17))
    h = c.add_arith_template(T)
    ins_all = []
    for _ in range(3):
        ins = c.new_wires(T.n_in)
        ins_all.append(ins)
        c.add_arith(h, ins)
    z = c.new_wire()
    c.add_constraint({z: 1}, {0: 1}, {})
    o = c.add_instance([z] * 17)
    c.add_constraint({o: 1}, {0: 1}, {1: 1}, defines=1)
    blob = c.pack()
    assert int(blob[0]) == R1.MAGIC2 and int(blob[11]) == 1 and int(blob[2]) == c.n_constraints == 613 + 2 + 3 * len(T.rows)
    vals = {0: 1, z: 0}
    for ins in ins_all:
        for k in ins[:6]:
            vals[k] = rnd.randrange(P)
        for k in ins[6:11]:
            vals[k] = rnd.randrange(2)
        for k in ins[11:19]:
            vals[k] = rnd.randrange(1 << 64)
        vals[ins[19]] = rnd.randrange(R)
    ref = c.complete(vals)
    ids = np.array(list(vals.keys()), dtype=np.int64)

    def arrays(v):
        w = np.zeros((c.n_wires, 4), dtype=np.uint64)
        mask = np.zeros(c.n_wires, dtype=np.uint8)
        w[ids] = native.fr_words([v[int(k)] for k in ids])
        mask[ids] = 1
        return w, mask
    wf, a, b_, cc = native.r1cs_eval(blob, *arrays(vals))
    assert native.fr_ints(wf) == ref
    rows = list(c.rows())
    dot = lambda M: sum(co * ref[k] for k, co in M.items()) % R
    for i in list(range(0, len(rows), 97)) + [len(rows) - 1]:
        A, B, C = rows[i]
        assert dot(A) * dot(B) % R == dot(C)
        assert native.fr_ints(a[i:i + 1])[0] == dot(A) and native.fr_ints(b_[i:i + 1])[0] == dot(B) and native.fr_ints(cc[i:i + 1])[0] == dot(C)
    # no witness: the inverse of zero in the second instance; a complete witness with one arithmetic wire changed
    v2 = dict(vals)
    for k in ins_all[1][:3]:
        v2[k] = 0
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(blob, *arrays(v2))
    full = np.ones(c.n_wires, dtype=np.uint8)
    base = c.ariths[0][1][2][1]
    for k in (base, base + T.n_int // 2, base + T.n_int - 1):
        w2 = wf.copy()
        w2[k] = native.fr_words([(native.fr_ints(w2[k:k + 1])[0] + 1) % R])[0]
        with pytest.raises(ValueError, match="does not satisfy"):
            native.r1cs_eval(blob, w2, full.copy())
    # a truncated or inconsistent blob is refused
    for bad in (blob[:-1], np.concatenate([blob[:11], blob[11:12] + np.uint64(1), blob[12:]])):
        with pytest.raises(native.ZpError):
            native.r1cs_eval(bad, *arrays(vals))
    # key scalars: the definition over every row (instances, explicit constraints, template rows) for every wire
    tau, al, be, ga, de = 12345, 2, 3, 5, 7
    u, v, l, hh = native.r1cs_key_scalars(blob, tau, al, be, ga, de)
    m = 1 << c.logm()
    om = pow(5, (R - 1) // m, R)
    zt = (pow(tau, m, R) - 1) % R
    Ls = [zt * pow(om, i, R) % R * pow(m * (tau - pow(om, i, R)) % R, -1, R) % R for i in range(len(rows))]
    uu, vv, ww = [0] * c.n_wires, [0] * c.n_wires, [0] * c.n_wires
    for i, (A, B, C) in enumerate(rows):
        for k, cf in A.items():
            uu[k] = (uu[k] + cf * Ls[i]) % R
        for k, cf in B.items():
            vv[k] = (vv[k] + cf * Ls[i]) % R
        for k, cf in C.items():
            ww[k] = (ww[k] + cf * Ls[i]) % R
    assert native.fr_ints(u) == uu and native.fr_ints(v) == vv
    assert native.fr_ints(l) == [(be * uu[j] + al * vv[j] + ww[j]) * pow(ga if j <= 1 else de, -1, R) % R for j in range(c.n_wires)]
