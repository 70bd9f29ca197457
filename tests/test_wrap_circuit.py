"""The Groth16 wrap's circuit and the R1CS machinery under it (CPU tests: everything here is host code of the library or Python).

* service/r1cs.py + csrc/r1cs.hip: the Poseidon-BN254 (t = 17) gadget template computes the permutation the oracle computes; zp_r1cs_eval
  completes a witness exactly as the Python reference does and gives A w, B w, C w per constraint; zp_r1cs_key_scalars matches the
  definition of a Groth16 key's scalars.
* service/wrap_circuit.py over a real final-STARK-shaped proof (BN128-hash mode, from the CPU checker's backend): the witness completes, the
  public input is what oracle/wrap_verify.py recomputes from the STARK on its own, a flipped Merkle digest / opened value / index has NO
  satisfying witness, and a Groth16 proof of it (the checker's trapdoor prover under the product's key scalars) passes the pairing check."""
import copy
import json

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
from eigen_zeth_amd.service import groth16 as G16
from eigen_zeth_amd.service import r1cs as R1
from eigen_zeth_amd.service import wrap_circuit as WC
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from oracle import groth16_verify as GV
from oracle import naive as NV
from oracle import stark_verify as V
from oracle import wrap_verify as WV
from cpu_wrap_backend import CpuWrapBackend as CpuBackend

R = R1.R


@pytest.fixture(scope="module")
def bn():
    return bn254_poseidon_params(17)


@pytest.fixture(scope="module")
def small():
    """two chained permutation gadgets + glue: pub = out2, a boolean wire"""
    c = R1.Circuit(R1.poseidon_template(17))
    ins = c.new_wires(17)
    o1 = c.add_instance(ins)
    more = c.new_wires(16)
    o2 = c.add_instance([o1] + more)
    b = c.new_wire()
    c.add_constraint({o2: 1}, {0: 1}, {1: 1}, defines=1)
    c.add_constraint({b: 1}, {b: 1}, {b: 1})
    vals = {ins[i]: i * 7 + 3 for i in range(17)}
    vals.update({more[i]: i * i + 11 for i in range(16)})
    vals[b] = 1
    return c, ins, more, o1, o2, b, vals


def _arrays(c, vals):
    w = np.zeros((c.n_wires, 4), dtype=np.uint64)
    mask = np.zeros(c.n_wires, dtype=np.uint8)
    for k, v in list(vals.items()) + [(0, 1)]:
        w[k] = native.fr_words([v])[0]
        mask[k] = 1
    return w, mask


def test_gadget_is_the_permutation_and_eval_matches_the_reference(small, bn):
    c, ins, more, o1, o2, b, vals = small
    rc, mds, rp = bn
    tpl = c.tpl
    assert len(tpl.cons) == 3 * (8 * 17 + 68) + 1 == 613 and tpl.n_local == 631
    d1 = NV.poseidon_bn254_perm([vals[w] for w in ins], rc, mds, rp)[0]
    d2 = NV.poseidon_bn254_perm([d1] + [vals[w] for w in more], rc, mds, rp)[0]
    full = c.complete(vals)
    assert full[o1] == d1 and full[o2] == d2 and full[1] == d2
    blob = c.pack()
    wf, a, bb, cc = native.r1cs_eval(blob, *_arrays(c, vals))
    assert native.fr_ints(wf) == full
    dot = lambda M: sum(cf * full[k] for k, cf in M.items()) % R
    ai, bi, ci = native.fr_ints(a), native.fr_ints(bb), native.fr_ints(cc)
    rows = list(c.rows())
    for q, (A, B, C) in enumerate(rows):
        assert ai[q] == dot(A) and bi[q] == dot(B) and ci[q] == dot(C) and ai[q] * bi[q] % R == ci[q], q
    assert not any(ai[len(rows):]) and len(ai) == 1 << c.logm()
    bad = dict(vals)
    bad[b] = 2                                                   # not a bit
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(blob, *_arrays(c, bad))
    w, mask = _arrays(c, vals)
    mask[more[3]] = 0                                            # a wire nobody set
    with pytest.raises(native.ZpError):
        native.r1cs_eval(blob, w, mask)
    with pytest.raises(native.ZpError):
        native.r1cs_eval(blob[:-3], *_arrays(c, vals))


def test_key_scalars_match_the_definition(small):
    c = small[0]
    blob = c.pack()
    tau, al, be, ga, de = 123456789, 5, 7, 11, 13
    u, v, l, h = native.r1cs_key_scalars(blob, tau, al, be, ga, de)
    m, om = 1 << c.logm(), pow(5, (R - 1) >> c.logm(), R)
    zt = (pow(tau, m, R) - 1) % R
    rows = list(c.rows())
    L = [zt * pow(om, i, R) % R * pow(m, -1, R) % R * pow((tau - pow(om, i, R)) % R, -1, R) % R for i in range(len(rows))]
    uu, vv, ww = [0] * c.n_wires, [0] * c.n_wires, [0] * c.n_wires
    for i, (A, B, C) in enumerate(rows):
        for k, cf in A.items():
            uu[k] = (uu[k] + cf * L[i]) % R
        for k, cf in B.items():
            vv[k] = (vv[k] + cf * L[i]) % R
        for k, cf in C.items():
            ww[k] = (ww[k] + cf * L[i]) % R
    assert native.fr_ints(u) == uu and native.fr_ints(v) == vv
    li = native.fr_ints(l)
    for j in range(c.n_wires):
        assert li[j] == (be * uu[j] + al * vv[j] + ww[j]) * pow(ga if j <= 1 else de, -1, R) % R
    hi = native.fr_ints(h)
    assert len(hi) == m - 1 and hi[0] == zt * pow(de, -1, R) % R and hi[7] == pow(tau, 7, R) * zt * pow(de, -1, R) % R
    with pytest.raises(native.ZpError):                           # tau on the domain is not a usable point
        native.r1cs_key_scalars(blob, 1, al, be, ga, de)


@pytest.fixture(scope="module")
def final_like(tables, bn):
    """a BN128-hash-mode STARK with the tree kinds a final STARK has (grouped trace / quotient leaves, two FRI layers of different widths)"""
    cpu = CpuBackend(*tables, hash_mode="bn128", bn_tables=bn)
    air = AIR.get_air("wide8")
    tr, pub = native.synth_trace(air.trace_kind, 6, air.width, 5)
    params = PR.StarkParams(6, 2, 2, 3, 3, pow_bits=0, hash="bn128")
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu)))
    return cpu, air, params, proof


def test_wrap_circuit_over_a_bn128_stark(final_like, tables, bn):
    cpu, air, params, proof = final_like
    rc, mds = tables
    assert V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()), bn)
    lay = WC.Layout.of_air(air, params)
    assert [t[0] for t in lay.trees] == ["trace", "quotient", "fri0", "fri1"]
    wc = WC.wrap_circuit(lay)
    aux = 479881985774944702531460751064278034642760119942
    # the transcript of the proof, replayed by the product's own sponge (its permutation = the checker's)
    head = WC.head_values(air, params, proof["root32"], proof["shift"])
    tlog = WC.TranscriptLog(proof, lay, head)
    assert WC.perm17(list(range(17))) == NV.poseidon_bn254_perm(list(range(17)), *bn)
    assert tlog.indices == [q["index"] for q in proof["queries"]] and tlog.data == WV.transcript_data(proof, air.program(), bn)
    assert len(tlog.blocks) == len(wc.tblocks) == 9 and len(tlog.rates) == 1
    w0, mask = wc.assign(proof, aux, tlog)
    wf, a, b, c = native.r1cs_eval(wc.blob, w0, mask)
    d = native.fr_ints(wf[1:2])[0]
    assert d == WV.public_input(proof, aux, bn, air.program()) != WV.public_input(proof, aux + 1, bn, air.program())
    # the library's assignment (zp_wrap_assign over the circuit's script and the binary openings) sets the same wires to the same values
    rec = WC.openings_record(proof, lay, tlog)
    set_idx, set_val = native.wrap_assign(wc.script, rec, aux)
    assert sorted(set_idx.tolist()) == np.flatnonzero(mask).tolist() and (w0[set_idx.astype(np.int64)] == set_val).all()
    for cut in (rec[:-1], rec[1:], np.concatenate([rec[:1], rec[1:2] + np.uint64(1), rec[2:]])):
        with pytest.raises(ValueError):
            native.wrap_assign(wc.script, cut, aux)
    with pytest.raises(ValueError):
        native.wrap_assign(wc.script[:-6], rec, aux)
    lay2 = WC.Layout.of_air(air, PR.StarkParams(6, 2, 2, 3, 4, pow_bits=0, hash="bn128"))
    with pytest.raises(ValueError):                                # the script of another layout
        native.wrap_assign(WC.wrap_circuit(lay2).script, rec, aux)
    # no satisfying witness for a STARK whose openings do not hash to its roots
    for mutate in (lambda p: p["queries"][1]["fri"][0]["path"][0].__setitem__(3, str((int(p["queries"][1]["fri"][0]["path"][0][3]) + 1) % R)),
                   lambda p: p["queries"][0]["trace"]["values"].__setitem__(2, p["queries"][0]["trace"]["values"][2] ^ 1),
                   lambda p: p["queries"][2].__setitem__("index", p["queries"][2]["index"] ^ 1),
                   lambda p: p["roots"]["quotient"].__setitem__(0, str((int(p["roots"]["quotient"][0]) + 1) % R))):
        bad = copy.deepcopy(proof)
        mutate(bad)
        with pytest.raises(ValueError, match="does not satisfy"):
            native.r1cs_eval(wc.blob, *wc.assign(bad, aux, tlog))
    # a Groth16 proof of the statement: the product's key (scalars by zp_r1cs_key_scalars), the checker's trapdoor prover, the pairing check
    key = G16.Key(wc.blob)
    proof_g, pubs, _ = G16.prove(key, set_idx, set_val, cpu, (11, 13))
    assert pubs == [d] and WV.verify(key.vk, proof_g, pubs, proof, aux, bn, air.program())
    assert not GV.verify(key.vk, proof_g, [(d + 1) % R])
    with pytest.raises(V.Reject):
        WV.verify(key.vk, proof_g, pubs, proof, aux + 1, bn, air.program())
    js = json.loads(G16.proof_to_json(proof_g))
    assert js["protocol"] == "groth16" and js["curve"] == "BN128" and js["pi_b"]["x"][0].isdigit()
    # what the holder of the final STARK still checks natively: the arithmetic, at the indices as given
    assert WV.verify_rest(proof, air.program(), rc, mds, V.expectation(params.to_dict()), bn)


def test_indices_are_bound_by_the_transcript_in_the_circuit(final_like, tables, bn):
    """stage B-1 (round 5): a final STARK whose query indices are ALTERED -- with openings and authentication paths that are consistent with the
    new positions, i.e. a proof round 4's circuit had a witness for -- has no witness any more: the bits that select the path positions are the
    transcript's.  Neither has one whose transcript data (an out-of-domain evaluation, the final layer, a parameter) differs from what was
    hashed.  The checker's side: with the arithmetic checked at the indices as given (verify_rest), only d and the pairing stand between an
    index swap and acceptance -- and d's circuit refuses it."""
    cpu, air, params, proof = final_like
    rc, mds = tables
    lay = WC.Layout.of_air(air, params)
    wc = WC.wrap_circuit(lay)
    aux = 7
    head = WC.head_values(air, params, proof["root32"], proof["shift"])
    tlog = WC.TranscriptLog(proof, lay, head)
    # (a) re-open the SAME commitments at another position: the prover of `proof` can do that for any index (it holds the trees) -- here the
    # openings of query 1 stand in for query 0 (valid paths, valid leaves, wrong place to look)
    moved = copy.deepcopy(proof)
    moved["queries"][0] = copy.deepcopy(proof["queries"][1])
    assert moved["queries"][0]["index"] != proof["queries"][0]["index"]
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(wc.blob, *wc.assign(moved, aux, tlog))
    # ... and claiming the rate elements that WOULD give those indices does not help: they are tied to the sponge's output state
    forged = copy.deepcopy(tlog)
    e = forged.rates[0][0]
    forged.rates[0][0] = (e & ~((1 << lay.logm) - 1)) | moved["queries"][0]["index"]
    forged.indices[0] = moved["queries"][0]["index"]
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(wc.blob, *wc.assign(moved, aux, forged))
    # (b) transcript data that is not what was absorbed
    for field in ("z", "final", "head"):
        bad, tl = copy.deepcopy(proof), None
        if field == "z":
            bad["evals"]["z"][3][1] ^= 1
            tl = WC.TranscriptLog(bad, lay, head)                  # an honest replay of the altered data: other indices come out
            assert tl.indices != tlog.indices
        elif field == "final":
            bad["fri"]["final"][2][5] ^= 1
            tl = WC.TranscriptLog(bad, lay, head)
        else:
            tl = WC.TranscriptLog(bad, lay, head[:6] + [head[6] + 1] + head[7:])     # another n_queries in the parameter block
        with pytest.raises(ValueError, match="does not satisfy"):
            native.r1cs_eval(wc.blob, *wc.assign(bad, aux, tl))      # the openings are those of the old indices
        # the data alone (indices kept as the old transcript gave them): the sponge's output no longer has those bits
        keep = copy.deepcopy(tl)
        keep.rates, keep.indices = tlog.rates, tlog.indices
        with pytest.raises(ValueError, match="does not satisfy"):
            native.r1cs_eval(wc.blob, *wc.assign(bad, aux, keep))
    # (c) a non-canonical decomposition (value + r) of a rate element is refused by the < r chain: flip the witness bits by hand
    w0, mask = wc.assign(proof, aux, tlog)
    bits = wc.ebits[(0, 0)]
    alt = tlog.rates[0][0] + R
    if alt < (1 << 254):
        for i, bw in enumerate(bits):
            w0[bw] = native.fr_words([(alt >> i) & 1])[0]
        with pytest.raises(ValueError, match="does not satisfy"):
            native.r1cs_eval(wc.blob, w0, mask)
    # the checker: d differs for the moved proof, so the honest Groth16 proof does not cover it
    assert WV.public_input(moved, aux, bn, air.program()) != WV.public_input(proof, aux, bn, air.program())
    # ... which is ALL that stands in its way: the arithmetic at the moved position is that of a genuine opening, so verify_rest (which takes the
    # indices as given) accepts it -- exactly why the indices had to be bound inside the circuit
    assert WV.verify_rest(moved, air.program(), rc, mds, V.expectation(params.to_dict()), bn)
    with pytest.raises(V.Reject):                                  # (the full verifier, which hashes the transcript itself, refuses it)
        V.verify(moved, air.program(), rc, mds, V.expectation(params.to_dict()), bn)


def test_stage_b2_the_circuit_runs_the_verifiers_arithmetic(final_like, tables, bn):
    """wrap stage B-2 (round 6): built FOR a statement, the circuit also runs the verifier's field arithmetic (service/wrap_arith.py) -- challenges off
    the in-circuit sponge, the constraint identity at zeta, the DEEP quotient and every fold at every query, the final layer's degree -- and its
    public input commits to PUBLIC data only.  An honest STARK has a witness (the library's host evaluator: witness programs + every row); the
    checker recomputes d from the statement, the public inputs, aux and zeta alone (no root, no evaluation, no opening); the library's assignment
    equals the reference assignment; every arithmetic wire is pinned; and a STARK made HONESTLY FROM A FALSE WITNESS -- every hash, every index and
    the whole transcript consistent, i.e. a proof the hashing-only circuit of rounds 4-5 has a witness for -- has none."""
    from eigen_zeth_amd.service import wrap_arith as WA
    cpu, air, params, proof = final_like
    lay = WC.Layout.of_air(air, params)
    head = WC.head_values(air, params, proof["root32"], proof["shift"])
    st = WA.Statement(air.program(), proof["root32"], proof["shift"], head)
    wc = WC.wrap_circuit(lay, st)
    assert int(wc.blob[0]) == R1.MAGIC2 and int(wc.blob[11]) == 2 and len(wc.c.ariths[1][1]) == params.n_queries
    tlog = WC.TranscriptLog(proof, lay, head)
    aux = [479881985774944702531460751064278034642760119942]       # + one element per sparse fixed column (wide8 has none)
    w0, mask = wc.assign(proof, aux, tlog)
    wf, a, b, c = native.r1cs_eval(wc.blob, w0, mask)
    d = native.fr_ints(wf[1:2])[0]
    zw = [(tlog.chal[1] >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
    args = (air.program(), params.to_dict(), proof["root32"], proof["shift"], proof["publics"])
    assert d == WV.public_input_b2(*args, aux[0], zw, bn)
    assert d != WV.public_input_b2(*args, aux[0] + 1, zw, bn) and d != WV.public_input_b2(*args, aux[0], [zw[0] ^ 1] + zw[1:], bn)
    assert d != WV.public_input_b2(air.program(), params.to_dict(), proof["root32"], proof["shift"], [proof["publics"][0] ^ 1] + proof["publics"][1:], aux[0], zw, bn)
    assert WC.zeta_of(tlog.chal[1]) == [v % V.P for v in zw]
    # the same wires from the library's assignment script over the binary openings record ("PZOPEN03": + the rate element behind every challenge)
    rec = WC.openings_record(proof, lay, tlog)
    set_idx, set_val = native.wrap_assign(wc.script, rec, aux)
    assert sorted(set_idx.tolist()) == np.flatnonzero(mask).tolist() and (w0[set_idx.astype(np.int64)] == set_val).all()
    aux_l, zw_l = native.wrap_aux(rec, air.program(), proof["publics"], params.logn, proof["root32"], aux[0])
    assert aux_l == aux and zw_l == zw
    # the reference completion (Python integers: Template.run + every row) gives the same witness
    vals = {int(k): v for k, v in zip(np.flatnonzero(mask), native.fr_ints(w0[np.flatnonzero(mask)]))}
    assert native.fr_ints(wf) == wc.c.complete(vals)
    # ... and the CHECKER's own reader of the blob (oracle/r1cs_blob.py: shares nothing with the builder or the library) finds every constraint --
    # Poseidon instances, explicit rows, arithmetic templates -- satisfied by that witness, and names the row a changed wire breaks
    from oracle import r1cs_blob as RB
    wi = native.fr_ints(wf)
    assert RB.first_violated(wc.blob, wi) == -1 and sum(1 for _ in RB.rows_of(wc.blob)) == wc.c.n_constraints
    k = wc.c.ariths[1][1][1][1] + 77
    assert RB.first_violated(wc.blob, wi[:k] + [(wi[k] + 1) % R] + wi[k + 1:]) >= len(wc.c.instances) * 613 + len(wc.c.extras)
    # every arithmetic wire is pinned by a row
    import random
    rnd = random.Random(3)
    full = np.ones(wc.c.n_wires, dtype=np.uint8)
    for tpl, insts in wc.c.ariths:
        for ins, base in (insts[0], insts[-1]):
            for _ in range(5):
                k = base + rnd.randrange(tpl.n_int)
                w2 = wf.copy()
                w2[k] = native.fr_words([(native.fr_ints(w2[k:k + 1])[0] + 1) % R])[0]
                with pytest.raises(ValueError, match="does not satisfy"):
                    native.r1cs_eval(wc.blob, w2, full.copy())
    # a FALSE witness, proven honestly: hashes, indices and transcript are all consistent -- the hashing-only circuit accepts, stage B-2 does not
    tr, pub = native.synth_trace(air.trace_kind, 6, air.width, 5)
    tr[3, 17] ^= np.uint64(1)
    bad = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu)))
    tl2 = WC.TranscriptLog(bad, lay, head)
    wc_hash_only = WC.wrap_circuit(lay)
    native.r1cs_eval(wc_hash_only.blob, *wc_hash_only.assign(bad, aux[0], tl2))
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(wc.blob, *wc.assign(bad, aux, tl2))
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), *tables, V.expectation(params.to_dict()), bn)
    # one wrong fold / one wrong opened value that still HASHES: impossible to stage without breaking a Merkle path, so the witness-level form --
    # the honest caller-set wires with one FRI leaf element's VALUE wire and its sponge left as they are, and the fold's expected value changed --
    # is what the pinned-wire loop above covers; the statement-level form is the false-witness proof.
    # Groth16 over the enlarged circuit: the product's key scalars, the checker's trapdoor prover, the pairing check, d from public data
    key = G16.Key(wc.blob)
    proof_g, pubs, _ = G16.prove(key, set_idx, set_val, cpu, (11, 13))
    assert pubs == [d] and WV.verify_b2(key.vk, proof_g, pubs, *args, aux[0], zw, bn)
    with pytest.raises(V.Reject):
        WV.verify_b2(key.vk, proof_g, pubs, *args, aux[0] + 1, zw, bn)
    assert "stage B-2" in G16.circuit_text(wc, key)
