"""CPU tests of the STARK stage logic (no GPU): the orchestration + the independent verifier on the
oracle backend, AIR code generation, transcript."""
import copy

import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from eigen_zeth_amd.stark.transcript import Transcript
from oracle import stark_verify as V
from oracle.stark_cpu import CpuBackend


@pytest.fixture(scope="module")
def be(tables):
    return CpuBackend(*tables)


def test_synthetic_traces_satisfy_their_airs():
    P = V.P
    tr, pub = native.synth_trace(0, 6, 2, 1)
    for i in range(63):
        assert int(tr[0, i + 1]) == int(tr[1, i]) and int(tr[1, i + 1]) == (int(tr[0, i]) + int(tr[1, i])) % P
    assert pub.tolist() == [int(tr[0, 0]), int(tr[1, 0]), int(tr[1, 63])]
    tr, pub = native.synth_trace(1, 5, 6, 2)
    for r in range(31):
        for i in range(6):
            assert int(tr[i, r + 1]) == (int(tr[i, r]) * int(tr[(i + 1) % 6, r]) + int(tr[(i + 2) % 6, r]) + i) % P
    with pytest.raises(ValueError):
        native.synth_trace(0, 6, 3, 1)


def test_codegen_is_deterministic_and_shares_subexpressions():
    a = AIR.wide_air(8)
    src = AIR.emit_quotient_source(a, "hip")
    assert src == AIR.emit_quotient_source(AIR.wide_air(8), "hip")
    assert a.symbol in src and "__global__" in src
    with pytest.raises(AssertionError):
        AIR.emit_quotient_source(a, "c")      # the product emits device code only; the checker interprets the blob
    assert src.count("cols[(u64)0 * sc + i]") == 1  # each column value is loaded once (row-window form: column stride sc, local row i)


def test_transcript_is_deterministic_and_order_sensitive(be):
    t1, t2, t3 = Transcript(be.poseidon_perm), Transcript(be.poseidon_perm), Transcript(be.poseidon_perm)
    t1.absorb([1, 2, 3]); t2.absorb([1, 2, 3]); t3.absorb([3, 2, 1])
    a, b, c = t1.squeeze(5), t2.squeeze(5), t3.squeeze(5)
    assert a == b and a != c
    assert t1.squeeze(9) == t2.squeeze(9)


@pytest.mark.parametrize("name,logn", [("fib", 5), ("wide8", 7), ("perm", 6), ("chunk16", 6)])
def test_cpu_proof_verifies_and_tampering_is_rejected(be, tables, name, logn):
    rc, mds = tables
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 77)
    params = PR.StarkParams(logn, 1, 3, 3, 6)
    proof = PR.prove(air, tr, pub, params, be)
    assert V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))
    for mutate in (lambda p: p["evals"]["zw"][0].__setitem__(1, 5),
                   lambda p: p["roots"]["trace"].__setitem__(0, 1),
                   lambda p: p["publics"].__setitem__(0, 7),
                   lambda p: p["queries"][0]["quotient"]["values"].__setitem__(2, 9),
                   lambda p: p["queries"][2]["fri"][0]["path"][0].__setitem__(0, 3)):
        bad = copy.deepcopy(proof)
        mutate(bad)
        with pytest.raises(V.Reject):
            V.verify(bad, air.program(), rc, mds, V.expectation(params.to_dict()))
    with pytest.raises(V.Reject):
        V.verify(proof, AIR.get_air("wide32").program(), rc, mds, V.expectation(params.to_dict()))


def test_wrong_witness_cannot_be_proven(be, tables):
    rc, mds = tables
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(0, 6, 2, 5)
    tr[0, 9] ^= 1
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, tr, pub, params, be)
    with pytest.raises(V.Reject):
        V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))


def test_permutation_argument_rejects_non_permutation(be, tables):
    rc, mds = tables
    air = AIR.get_air("perm")
    tr, pub = native.synth_trace(2, 6, 3, 11)
    assert sorted(tr[0].tolist()) == sorted(tr[1].tolist())
    tr[1, 5] = (int(tr[1, 5]) + 1) % V.P
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, tr, pub, params, be)
    with pytest.raises(V.Reject):
        V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))
    good, pub = native.synth_trace(2, 6, 3, 11)
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, good, pub, params, be)
    bad = copy.deepcopy(proof)
    bad["queries"][0]["stage2"]["values"][0] ^= 1
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, V.expectation(params.to_dict()))
    bad = copy.deepcopy(proof)
    del bad["roots"]["stage2"]
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, V.expectation(params.to_dict()))


def test_chunk_trace_satisfies_every_constraint_on_the_trace_domain(tables):
    """row-by-row evaluation of the chunk AIR (wide mix + Fibonacci + permutation + LogUp range check) on the
    synthetic witness with its stage-2 columns from the oracle: every constraint vanishes on every row"""
    from oracle import oracle as O
    import numpy as np
    P = V.P
    logn, N = 5, 32
    from oracle.air_program import Program
    air = AIR.get_air("chunk16")
    prog = Program(air.program())
    assert prog.digest() == air.digest() and prog.n_constraints == len(air.constraints)
    tr, pub = native.synth_trace(3, logn, 16, 9)
    Ww = 8
    assert int(pub[7]) == N - 1 and sorted(tr[Ww + 2].tolist()) == sorted(tr[Ww + 3].tolist())
    assert int(tr[Ww + 5].sum()) == N and all(int(v) < N for v in tr[Ww + 2])
    g = [12345, 678, 91011]
    s2 = np.concatenate([O.grand_product(tr[Ww + 2], tr[Ww + 3], g), O.logup_columns(tr[Ww + 2], tr[Ww + 4], tr[Ww + 5], g)])
    full = np.concatenate([tr, s2])
    assert full.shape[0] == air.width + air.width2 == 28
    w = O.lib().orc_root(O.ROOT32_DEFAULT, logn)
    wlast = pow(w, N - 1, P)
    for i in range(N):
        cur = [int(v) for v in full[:, i]]
        nxt = [int(v) for v in full[:, (i + 1) % N]]
        fixed = [1 if i == 0 else 0, 1 if i == N - 1 else 0]
        vals = prog.evaluate_base(cur, nxt, fixed, [int(v) for v in pub] + g, (pow(w, i, P) - wlast) % P)
        assert all(v == 0 for v in vals), (i, [k for k, v in enumerate(vals) if v])


def test_lookup_argument_rejects_out_of_range_value(be, tables):
    rc, mds = tables
    air = AIR.get_air("chunk16")
    tr, pub = native.synth_trace(3, 6, 16, 13)
    Ww = 8
    bad = tr.copy()
    # a value outside [0, 2^k): keep c = r^2, d = fa*r + fb and the permutation consistent so that only the lookup fails
    j = 7
    bad[Ww + 2, j] = 1 << 20
    bad[Ww + 3, :] = bad[Ww + 2, [(5 * i + 3) % 64 for i in range(64)]]
    bad[Ww + 6, j] = (int(bad[Ww + 2, j]) ** 2) % V.P
    bad[Ww + 7, j] = (int(bad[Ww, j]) * int(bad[Ww + 2, j]) + int(bad[Ww + 1, j])) % V.P
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, bad, pub, params, be)
    with pytest.raises(V.Reject):
        V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, tr, pub, params, be)
    assert V.verify(proof, air.program(), rc, mds, V.expectation(params.to_dict()))
    assert len(proof["queries"][0]["stage2"]["values"]) == 12


def test_verifier_uses_its_own_parameters_not_the_proofs(be, tables):
    """a proof that claims weaker parameters (fewer or zero queries) must not verify: the verifier's parameter set is
    the caller's, and every parameter is bound into the transcript"""
    rc, mds = tables
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(0, 6, 2, 5)
    params = PR.StarkParams(6, 1, 3, 3, 6)
    proof = PR.prove(air, tr, pub, params, be)
    exp = V.expectation(params.to_dict())
    assert V.verify(proof, air.program(), rc, mds, exp)
    # forged: zero queries claimed, query list emptied, evaluations then unconstrained
    bad = copy.deepcopy(proof)
    bad["params"]["n_queries"] = 0
    bad["queries"] = []
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, exp)
    with pytest.raises(V.Reject):      # ... also when the verifier is (wrongly) configured with zero queries
        V.verify(bad, air.program(), rc, mds, dict(exp, n_queries=0))
    # a proof made with other parameters does not verify under the caller's
    weak = PR.prove(air, tr, pub, PR.StarkParams(6, 1, 3, 3, 2), be)
    with pytest.raises(V.Reject):
        V.verify(weak, air.program(), rc, mds, exp)
    relabel = copy.deepcopy(weak)
    relabel["params"]["n_queries"] = 6     # same proof relabelled: transcript binding breaks it
    with pytest.raises(V.Reject):
        V.verify(relabel, air.program(), rc, mds, exp)
    for key, val in (("root32", 7277203076849721926), ("shift", 7)):
        with pytest.raises(V.Reject):
            V.verify(proof, air.program(), rc, mds, dict(exp, **{key: val}))
    with pytest.raises(ValueError):
        V.verify(proof, air.program(), rc, mds, {"logn": 6})


def test_proof_of_work_is_checked(be, tables):
    rc, mds = tables
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(0, 6, 2, 5)
    params = PR.StarkParams(6, 1, 3, 3, 6, pow_bits=10)
    assert params.security_bits() == 16
    proof = PR.prove(air, tr, pub, params, be)
    exp = V.expectation(params.to_dict())
    assert "pow_nonce" in proof and V.verify(proof, air.program(), rc, mds, exp)
    bad = copy.deepcopy(proof)
    bad["pow_nonce"] += 1
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, exp)
    del bad["pow_nonce"]
    with pytest.raises(V.Reject):
        V.verify(bad, air.program(), rc, mds, exp)


def test_constraint_program_blob_is_checked_and_statement_bound(tables):
    import numpy as np
    from oracle.air_program import Program, BadProgram
    blob = AIR.get_air("chunk16").program()
    pr = Program(blob)
    assert (pr.width, pr.width2, pr.n_pub, pr.n_chal, pr.q_chunks) == (16, 12, 8, 3, 1) and pr.n_slots <= 16
    for mutate in (lambda b: b.__setitem__(0, 1), lambda b: b.__setitem__(7, int(b[7]) + 1),
                   lambda b: b.__setitem__(12 + int(b[6]), 9)):      # magic, length, opcode
        bad = blob.copy()
        mutate(bad)
        with pytest.raises(BadProgram):
            Program(bad)
    other = blob.copy()
    other[12] = (int(other[12]) + 1) % V.P            # one constant changed: another statement, another digest
    assert Program(other).digest() != pr.digest()


def test_degree_three_constraints_split_the_quotient(be, tables):
    """constraints of degree 3 (+1 for the transition factor): the quotient has degree < 2N and is committed as two pieces of
    degree < N (6 base columns); q(z) = q0(z) + (z/shift)^N q1(z) is what the verifier checks"""
    rc, mds = tables
    air = AIR.get_air("cubic")
    assert AIR.quotient_chunks(air) == 2 and AIR.quotient_chunks(AIR.get_air("chunk16")) == 1
    tr, pub = AIR.cubic_witness(6, 3)
    params = PR.StarkParams(6, 1, 2, 3, 6, pow_bits=4)
    proof = PR.prove(air, tr, pub, params, be)
    exp = V.expectation(params.to_dict())
    assert len(proof["evals"]["z"]) == 2 + 6 and len(proof["queries"][0]["quotient"]["values"]) == 6
    assert V.verify(proof, air.program(), rc, mds, exp)
    for mutate in (lambda p: p["evals"]["z"][5].__setitem__(0, 1), lambda p: p["queries"][1]["quotient"]["values"].__setitem__(4, 2)):
        bad = copy.deepcopy(proof)
        mutate(bad)
        with pytest.raises(V.Reject):
            V.verify(bad, air.program(), rc, mds, exp)
    tr[1, 9] ^= 1
    with pytest.raises(V.Reject):
        V.verify(PR.prove(air, tr, pub, params, be), air.program(), rc, mds, exp)
    with pytest.raises(AssertionError):          # blow-up 1 cannot hold a degree-2N quotient
        PR.prove(air, tr, pub, PR.StarkParams(6, 0, 2, 3, 6), be)
