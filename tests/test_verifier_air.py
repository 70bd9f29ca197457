"""The STARK-verifier AIR (eigen_zeth_amd/stark/verifier_air.py): GenAggregatedProof and the final STARK prove what they name -- everything a
verifier of the inner proofs does at their queries: Merkle paths, DEEP quotient, FRI folds, with the opened values PRIVATE -- plus their
Fiat-Shamir transcripts (proto/prover/v1/prover.proto:115-148; client src/prover/provider.rs:422-503).  CPU tests: the product's AIR /
witness builder / orchestration over the checker's backend, judged by the checker's independent verifiers (oracle/stark_verify.py,
oracle/aggregate_verify.py)."""
import copy
import json

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from eigen_zeth_amd.stark import verifier_air as VA
from oracle import aggregate_verify as AV
from oracle import oracle as O
from oracle import stark_verify as V
from oracle.stark_cpu import CpuBackend

P = O.P


def strip_paths(proof):
    """an inner proof as an aggregated proof carries it: its header -- no query openings at all"""
    return {k: copy.deepcopy(v) for k, v in proof.items() if k != "queries"}


@pytest.fixture(scope="module")
def cpu(tables):
    return CpuBackend(*tables)


@pytest.fixture(scope="module")
def inner(cpu):
    """two chunk16 proofs (trace + stage-2 + quotient + two FRI layers: every kind of tree, leaves of 16 / 12 / 3 / 12 / 6 values)"""
    air = AIR.get_air("chunk16")
    params = PR.StarkParams(6, 1, 2, 3, 4, pow_bits=4)
    proofs = []
    for seed in (3, 4):
        tr, pub = native.synth_trace(air.trace_kind, 6, air.width, seed)
        proofs.append(json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu))))
    return air, params, proofs


@pytest.fixture(scope="module")
def aggregated(cpu, inner, tables):
    air, params, proofs = inner
    shape = VA.Shape.of_proof(proofs[0], 2)
    vair = VA.verifier_air(shape, *tables)
    trace, pubs = VA.build_witness(shape, proofs, cpu, air.digest_words())
    ap = VA.aggregation_params(shape, n_queries=5, fri_final_log=3)
    stark = PR.prove(vair, trace, pubs, ap, cpu)
    return shape, vair, ap, trace, pubs, {"kind": "aggregated", "inner": [strip_paths(p) for p in proofs], "stark": stark}


def test_poseidon_trace_rows_are_the_round_states(tables):
    rc, mds = tables
    x = O.random_field((5, 12), 77)
    st, cu = O.poseidon_trace(x, rc, mds)
    assert st.shape == (12, 160) and (st[:, 0::32].T == x).all()
    out = O.poseidon_perm(x, rc, mds)
    assert (st[:, 30::32].T == out).all() and (st[:, 31::32].T == out).all()
    for r in (0, 3, 4, 17, 29):                      # cubes: (state + round constant)^3
        for e in (0, 5, 11):
            v = (int(st[e, 32 + r]) + int(rc[12 * r + e])) % P
            assert int(cu[e, 32 + r]) == pow(v, 3, P)
    assert int(cu[2, 31]) == pow(int(st[2, 31]), 3, P)


def test_layout_and_schedule(inner):
    _, _, proofs = inner
    shape = VA.Shape.of_proof(proofs[0], 2)
    assert [t[0] for t in shape.trees] == ["trace", "stage2", "quotient", "fri0", "fri1"]
    k, periods, pb = shape.layout()
    assert periods & (periods - 1) == 0 and pb & (pb - 1) == 0 and k * periods >= shape.n_queries
    assert k * 2 * shape.blocks_per_proof() + 2 * len(shape.transcript_perms()) <= pb
    sched = shape.period_schedule()
    assert len(sched) == pb and sum(b["kind"] != "idle" for b in sched) == k * 2 * shape.blocks_per_proof()
    assert sum(b["last"] for b in sched) == k * 2 * len(shape.trees)      # one root comparison per (slot, proof, tree)
    # the service's shapes: two 2^20-row chunk proofs fit a 2^20-row verifier trace, its own proof a 2^18-row final trace
    big = VA.Shape(20, 1, 64, 12, 3, 80, 3, 5, 2, 16, 20)
    assert big.logn_trace() == 20
    assert VA.Shape(20, 2, 26, 0, 9, 50, 3, 5, 1, big.n_pub(), 0).logn_trace() == 18


def test_aggregated_proof_is_accepted_by_the_independent_verifier(inner, aggregated, tables):
    rc, mds = tables
    air, params, proofs = inner
    shape, vair, ap, _, pubs, agg = aggregated
    assert len(pubs) == shape.n_pub() and AIR.quotient_chunks(vair) == 3
    assert V.verify(agg["stark"], vair.program(), rc, mds, V.expectation(ap.to_dict()))
    assert AV.verify(agg, air.program(), vair.program(), rc, mds, V.expectation(params.to_dict()), V.expectation(ap.to_dict()), shape.n_slots())


def test_tampered_inner_proof_has_no_accepting_witness(inner, cpu):
    air, _, proofs = inner
    shape = VA.Shape.of_proof(proofs[0], 2)
    for mutate in (lambda p: p[1]["queries"][2]["trace"]["values"].__setitem__(5, p[1]["queries"][2]["trace"]["values"][5] ^ 1),
                   lambda p: p[0]["queries"][1]["fri"][0]["path"][2].__setitem__(1, p[0]["queries"][1]["fri"][0]["path"][2][1] ^ 1),
                   lambda p: p[0]["queries"][0]["stage2"]["path"][0].__setitem__(0, (p[0]["queries"][0]["stage2"]["path"][0][0] + 1) % P),
                   lambda p: p[1]["roots"]["quotient"].__setitem__(3, (p[1]["roots"]["quotient"][3] + 1) % P),
                   lambda p: p[0]["queries"][3].__setitem__("index", p[0]["queries"][3]["index"] ^ 2)):
        bad = copy.deepcopy(proofs)
        mutate(bad)
        with pytest.raises(ValueError, match="no accepting witness"):
            VA.build_witness(shape, bad, cpu, air.digest_words())


def test_forged_witnesses_are_rejected(inner, aggregated, cpu, tables):
    """a prover that skips the witness builder's own check cannot get a proof accepted: every way of bending the trace or the
    public inputs breaks a constraint (the out-of-domain identity fails in the independent verifier)"""
    rc, mds = tables
    shape, vair, ap, trace, pubs, _ = aggregated
    exp = V.expectation(ap.to_dict())
    prog = vair.program()

    def rejected(tr, pb):
        with pytest.raises(V.Reject):
            V.verify(PR.prove(vair, tr, pb, ap, cpu), prog, rc, mds, exp)
    t1 = trace.copy()
    t1[VA.S0 + 3, 32 * 7 + 11] = (int(t1[VA.S0 + 3, 32 * 7 + 11]) + 1) % P            # one state cell inside a permutation
    rejected(t1, pubs)
    p2 = pubs.copy()
    p2[shape.pub_root(1, 0, 2)] = (int(p2[shape.pub_root(1, 0, 2)]) + 1) % P          # claim another trace root for proof 1
    rejected(trace, p2)
    p3 = pubs.copy()
    p3[shape.pub_index(1, 0, 0)] ^= 1                                                 # claim another leaf index
    rejected(trace, p3)
    # swap left and right at one node: flip the direction bit (and the accumulated index) but keep the hashes
    sched = shape.period_schedule()
    b = next(i for i, blk in enumerate(sched) if blk["kind"] == "node" and blk["level"] == 2)
    t4 = trace.copy()
    t4[VA.COL_D, 32 * b:32 * b + 32] ^= 1
    rejected(t4, pubs)
    t5 = trace.copy()
    t5[VA.U0, 5] = (int(t5[VA.U0, 5]) + 1) % P                                        # a wrong cube
    rejected(t5, pubs)
    # ---- the arithmetic on the private openings
    p6 = pubs.copy()
    p6[shape.pub_arith(1, VA.AP_EZA)] = (int(p6[shape.pub_arith(1, VA.AP_EZA)]) + 1) % P   # another E_z: the DEEP value no longer matches layer 0
    rejected(trace, p6)
    p7 = pubs.copy()
    p7[shape.pub_finv(2, 0, 1)] = (int(p7[shape.pub_finv(2, 0, 1)]) + 1) % P               # another final-layer value than the last fold gives
    rejected(trace, p7)
    p12 = pubs.copy()
    p12[shape.pub_arith(0, shape.ap_beta(0, 1, 0))] = (int(p12[shape.pub_arith(0, shape.ap_beta(0, 1, 0))]) + 1) % P      # fold at another beta
    rejected(trace, p12)
    p13 = pubs.copy()
    p13[shape.pub_arith(0, VA.AP_G + 3)] = (int(p13[shape.pub_arith(0, VA.AP_G + 3)]) + 1) % P    # a wrong power of 1 / gamma
    rejected(trace, p13)
    fri_abs = next(i for i, blk in enumerate(sched) if blk["kind"] == "absorb" and blk["t"] == shape.t_quot + 1)
    for col, row in ((VA.HR0 + 2, 32 * fri_abs + 3),         # a register that is not the hashed value
                     (VA.ACA0, 32 * fri_abs + 1),             # a partial sum of the interpolation
                     (VA.ACB0 + 1, 32 * fri_abs + 2),         # ... of the fold
                     (VA.COL_TPX, 32 * fri_abs + 1),          # a power of 1 / x
                     (VA.COL_TAU, 32 * fri_abs + 5),          # tau changed in the middle of a layer
                     (VA.COL_X, 32 * (fri_abs + 2) + 7)):     # the evaluation point changed inside a block
        tt = trace.copy()
        tt[col, row] = (int(tt[col, row]) + 1) % P
        rejected(tt, pubs)
    # an opened value changed TOGETHER with every hash above it would need a second preimage; changed alone (registers and state of its
    # absorb block, rest of the permutation untouched) it breaks the round constraints
    t14 = trace.copy()
    t14[VA.S0 + 1, 32 * fri_abs] = (int(t14[VA.S0 + 1, 32 * fri_abs]) + 1) % P
    t14[VA.HR0 + 1, 32 * fri_abs:32 * fri_abs + 32] = t14[VA.S0 + 1, 32 * fri_abs]
    rejected(t14, pubs)
    assert len(pubs) > PR.PUBLICS_INLINE                                              # the publics enter the transcript through their digest
    # the transcripts: a claimed challenge that the sponge does not give, an absorbed value other than the hashed one, a cell of
    # a transcript permutation, a broken capacity chain
    perms = shape.transcript_perms()
    j_out = next(j for j, pm in enumerate(perms) if pm["out"] and not pm["pow"])
    p8 = pubs.copy()
    p8[shape.pub_tout(1, j_out, 2)] = (int(p8[shape.pub_tout(1, j_out, 2)]) + 1) % P
    rejected(trace, p8)
    p9 = pubs.copy()
    p9[shape.pub_tin(0, 1, 0)] = (int(p9[shape.pub_tin(0, 1, 0)]) + 1) % P
    rejected(trace, p9)
    blk = shape.transcript_block0() + len(perms) + 2                                  # proof 1, its third permutation
    t10 = trace.copy()
    t10[VA.S0 + 9, 32 * blk] = (int(t10[VA.S0 + 9, 32 * blk]) + 1) % P                # its capacity is not the previous permutation's
    rejected(t10, pubs)
    j_pow = len(perms) - 1
    assert perms[j_pow]["pow"]
    p11 = pubs.copy()
    p11[shape.pub_tout(0, j_pow, 0)] = 1                                              # claim a grinding digest with more leading zeros
    rejected(trace, p11)


def test_outer_publics_must_follow_the_inner_transcripts(inner, aggregated, tables):
    rc, mds = tables
    air, params, _ = inner
    shape, vair, ap, _, _, agg = aggregated
    args = (air.program(), vair.program(), rc, mds, V.expectation(params.to_dict()), V.expectation(ap.to_dict()), shape.n_slots())
    bad = copy.deepcopy(agg)
    bad["inner"][0]["evals"]["z"][0][0] ^= 1                   # a header that does not verify on its own
    with pytest.raises(V.Reject):
        AV.verify(bad, *args)
    bad = copy.deepcopy(agg)
    bad["inner"][1]["fri"]["final"][0][0] ^= 1                 # changes the transcript after the roots: other indices
    with pytest.raises(V.Reject):
        AV.verify(bad, *args)
    bad = copy.deepcopy(agg)
    bad["inner"] = bad["inner"][::-1]                          # the outer proof names proof 0's roots first
    with pytest.raises(V.Reject, match="public inputs|public transcript"):
        AV.verify(bad, *args)
    bad = copy.deepcopy(agg)
    bad["inner"][0]["queries"] = []                            # an aggregated proof carries no openings
    with pytest.raises(V.Reject, match="no openings"):
        AV.verify(bad, *args)
    assert all("queries" not in h for h in agg["inner"])
    bad = copy.deepcopy(agg)
    bad["inner"][1]["evals"]["zw"][3][2] ^= 1                  # changes E_zw (and the transcript): the arithmetic publics no longer follow
    with pytest.raises(V.Reject):
        AV.verify(bad, *args)
    bad = copy.deepcopy(agg)
    bad["inner"][0]["pow_nonce"] ^= 1                          # another nonce than the one the circuit hashed
    with pytest.raises(V.Reject, match="transcript|grinding"):
        AV.verify(bad, *args)
    bad = copy.deepcopy(agg)                                   # a transcript section that is too long / too short
    bad["stark"]["publics"] = list(bad["stark"]["publics"]) + [0]
    with pytest.raises(V.Reject):
        AV.verify(bad, *args)
    with pytest.raises(V.Reject):                              # fewer inner queries than the verifier requires
        AV.verify(agg, air.program(), vair.program(), rc, mds, dict(V.expectation(params.to_dict()), n_queries=5), V.expectation(ap.to_dict()),
                  shape.n_slots())


def test_single_proof_shape_and_identity_leaves(cpu, tables):
    """n_proofs = 1 (what the final STARK verifies: the aggregated proof's own STARK) and a tree whose leaves are not hashed
    (a quotient of one piece: 3 values, identity-padded) right after an idle wrap-around"""
    rc, mds = tables
    air = AIR.get_air("fib")
    params = PR.StarkParams(5, 1, 2, 3, 3, pow_bits=0)
    tr, pub = native.synth_trace(air.trace_kind, 5, air.width, 9)
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu)))
    shape = VA.Shape.of_proof(proof, 1)
    assert shape.trees[0][1] == 2 and VA.Shape.absorb_blocks(2) == 0             # the trace leaves of `fib` are identity leaves too
    vair = VA.verifier_air(shape, rc, mds)
    trace, pubs = VA.build_witness(shape, [proof], cpu, air.digest_words())
    ap = VA.aggregation_params(shape, n_queries=4, fri_final_log=3)
    assert V.verify(PR.prove(vair, trace, pubs, ap, cpu), vair.program(), rc, mds, V.expectation(ap.to_dict()))
    assert [int(v) for v in pubs][:shape.merkle_pubs()] == VA.expected_publics(shape, [proof])


def test_untrusted_shapes_and_parameters_are_bounded(inner):
    """what a recursive-proof text could choose freely sizes nothing: an aggregated proof's shape dictionary and a chunk proof's
    parameters are range-checked before any schedule, table or array is built from them (service/engine.py)"""
    from eigen_zeth_amd.service.engine import Engine
    _, _, proofs = inner
    shape = VA.Shape.of_proof(proofs[0], 2)
    assert VA.Shape.from_dict(shape.to_dict()).key() == shape.key()
    for k, v in (("logn", 99), ("W", -1), ("n_queries", True), ("n_proofs", 0), ("n_pub_inner", 1 << 40), ("logb", "1")):
        with pytest.raises(ValueError):
            VA.Shape.from_dict({**shape.to_dict(), k: v})
    with pytest.raises(KeyError):
        VA.Shape.from_dict({k: v for k, v in shape.to_dict().items() if k != "pow_bits"})
    text = json.dumps(proofs[0])
    assert Engine._parse_and_prepare(text)[2]["index"].shape == (shape.n_queries,)
    for k, v in (("logn", 1 << 30), ("n_queries", 10 ** 9), ("logb", 0), ("fri_logf", -3), ("pow_bits", 1.5)):
        bad = copy.deepcopy(proofs[0])
        bad["params"][k] = v
        with pytest.raises(ValueError, match="out of range"):
            Engine._parse_and_prepare(json.dumps(bad))
    bad = copy.deepcopy(proofs[0])
    bad["queries"] = bad["queries"][:-1]                     # fewer openings than the parameters promise
    with pytest.raises(ValueError):
        Engine._parse_and_prepare(json.dumps(bad))
    bad = copy.deepcopy(proofs[0])
    bad["queries"][1]["trace"]["path"][0] = bad["queries"][1]["trace"]["path"][0][:3]      # a ragged authentication path
    with pytest.raises(ValueError, match="shape"):
        Engine._parse_and_prepare(json.dumps(bad))


def test_aggregated_proofs_fold_again(inner, aggregated, cpu, tables):
    """recursion one level up: two aggregated proofs (each over two chunk proofs) are folded through their aggregation STARKs; the
    checker walks the tree -- outer STARK + native arithmetic for the two aggregation STARKs, then, for each of them, native
    arithmetic for ITS chunk proofs against its (now vouched-for) public inputs.  A chunk proof tampered at a leaf, a child
    attached to the wrong parent, or a child STARK whose publics were edited are all refused."""
    rc, mds = tables
    air, params, proofs = inner
    shape1, vair1, ap1, _, _, aggA = aggregated
    more = []
    for seed in (5, 6):
        tr, pub = native.synth_trace(air.trace_kind, 6, air.width, seed)
        more.append(json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, cpu))))
    trB, pubsB = VA.build_witness(shape1, more, cpu, air.digest_words())
    aggB = {"shape": shape1.to_dict(), "inner": [strip_paths(p) for p in more], "stark": PR.prove(vair1, trB, pubsB, ap1, cpu)}
    A = {"shape": shape1.to_dict(), "inner": aggA["inner"], "stark": json.loads(PR.proof_to_json(aggA["stark"]))}
    B = dict(aggB, stark=json.loads(PR.proof_to_json(aggB["stark"])))
    shape2 = VA.Shape.of_proof(A["stark"], 2)
    vair2 = VA.verifier_air(shape2, rc, mds)
    tr2, pubs2 = VA.build_witness(shape2, [A["stark"], B["stark"]], cpu, vair1.digest_words())
    ap2 = VA.aggregation_params(shape2, n_queries=4, fri_final_log=3)
    top = {"shape": shape2.to_dict(), "level": 2, "inner": [strip_paths(A["stark"]), strip_paths(B["stark"])],
           "children": [{"shape": A["shape"], "inner": A["inner"]}, {"shape": B["shape"], "inner": B["inner"]}],
           "stark": json.loads(PR.proof_to_json(PR.prove(vair2, tr2, pubs2, ap2, cpu)))}

    def program_of_shape(d):
        sh = VA.Shape.from_dict(d)
        return VA.verifier_air(sh, rc, mds).program(), sh.n_slots()

    def expect_of_shape(d):
        return V.expectation((ap1 if d == shape1.to_dict() else ap2).to_dict())
    args = (air.program(), V.expectation(params.to_dict()), rc, mds, program_of_shape, expect_of_shape)
    assert AV.verify_tree(top, *args)
    assert AV.verify_tree(dict(A), *args)                                   # a level-1 proof through the same entry
    bad = copy.deepcopy(top)
    bad["children"][1]["inner"][0]["evals"]["z"][2][1] ^= 1                  # a chunk proof at a leaf
    with pytest.raises(V.Reject):
        AV.verify_tree(bad, *args)
    bad = copy.deepcopy(top)
    bad["children"] = bad["children"][::-1]                                  # children under the wrong parents
    with pytest.raises(V.Reject):
        AV.verify_tree(bad, *args)
    bad = copy.deepcopy(top)
    bad["inner"][0]["publics"][3] = (bad["inner"][0]["publics"][3] + 1) % P  # an aggregation STARK claiming other public inputs
    with pytest.raises(V.Reject):
        AV.verify_tree(bad, *args)
    bad = copy.deepcopy(top)
    del bad["children"]                                                      # the chunk proofs withheld
    with pytest.raises((V.Reject, KeyError, ValueError)):
        AV.verify_tree(bad, *args)


@pytest.mark.parametrize("case", ["chunk16", "fib-single"])
def test_native_arithmetic_builder_equals_the_reference_walk(cpu, tables, inner, case):
    """zp_verifier_arith_host (csrc/recursion.hip: host C++ walk over the schedule descriptor + expansion; needs no GPU) writes the same 21
    columns as the readable reference (verifier_air.arith_columns), for mixed fold factors, stage-2 trees, unhashed leaves and a
    one-proof shape; inconsistent openings are refused by both"""
    if case == "chunk16":
        air, _, proofs = inner
    else:
        air = AIR.get_air("fib")
        tr, pub = native.synth_trace(air.trace_kind, 5, air.width, 9)
        proofs = [json.loads(PR.proof_to_json(PR.prove(air, tr, pub, PR.StarkParams(5, 1, 2, 3, 3, pow_bits=0), cpu)))]
    shape = VA.Shape.of_proof(proofs[0], len(proofs))
    keep = {}
    trace, _ = VA.build_witness(shape, proofs, cpu, air.digest_words(), keep=keep)
    got = native.verifier_arith_host(VA.arith_descriptor(shape), keep["arith_in"], threads=3)
    assert got.shape == (VA.WIDTH - VA.HR0, trace.shape[1]) and (got == trace[VA.HR0:]).all()
    assert (native.verifier_arith_host(VA.arith_descriptor(shape), keep["arith_in"], threads=1) == got).all()
    # a final-layer value that the last fold does not give / an opened value of a FRI leaf that the interpolation does not reproduce
    bad = dict(keep["arith_in"], fin=keep["arith_in"]["fin"].copy())
    bad["fin"][1, 0, 2] ^= np.uint64(1)
    with pytest.raises(ValueError, match="no accepting witness"):
        native.verifier_arith_host(VA.arith_descriptor(shape), bad)
    with pytest.raises(ValueError, match="no accepting witness"):
        VA.arith_columns(shape, bad)
    ops = VA._opening_table(shape)
    o = int(ops["sel"][(0, shape.t_quot + 1)][1])
    bad = dict(keep["arith_in"], vals=keep["arith_in"]["vals"].copy())
    bad["vals"][o, 1] = (int(bad["vals"][o, 1]) + 1) % P
    with pytest.raises(ValueError, match="no accepting witness"):
        native.verifier_arith_host(VA.arith_descriptor(shape), bad)
    with pytest.raises(ValueError, match="no accepting witness"):
        VA.arith_columns(shape, bad)
    with pytest.raises(native.ZpError):
        native.verifier_arith_host(VA.arith_descriptor(shape)[:-1], keep["arith_in"])


def test_native_proof_text_parser_equals_the_json_path(inner, cpu, tables, monkeypatch):
    """csrc/proofparse.hip: the openings of a proof text as arrays -- the same arrays prepare_proof() makes from json.loads, for the C++ writer's
    and the Python writer's spelling (with and without whitespace), with and without a stage-2 tree; anything else is refused (None: the
    caller's JSON path then words the error); the engine's fast path gives the same aggregated proof as its JSON path"""
    from eigen_zeth_amd.service.engine import Engine
    air, params, proofs = inner
    fib = AIR.get_air("fib")
    tr, pub = native.synth_trace(fib.trace_kind, 5, fib.width, 9)
    fibp = json.loads(PR.proof_to_json(PR.prove(fib, tr, pub, PR.StarkParams(5, 1, 2, 3, 3, pow_bits=0), cpu)))
    for pr in (proofs[0], proofs[1], fibp):
        want = VA.prepare_proof(pr)
        for text in (PR.proof_to_json(pr), json.dumps(pr, indent=1), json.dumps(dict(pr, chunk={"block": 1, "queries": "decoy"}))):
            got = VA.prepare_proof_text(text)
            assert got is not None
            obj, arr = got
            prep = VA.prepared_from_arrays(obj, arr)
            assert (prep["index"] == want["index"]).all() and prep["key"] == want["key"]
            assert all((a == b).all() for a, b in zip(prep["values"], want["values"])) and all((a == b).all() for a, b in zip(prep["paths"], want["paths"]))
            assert {k: v for k, v in obj.items() if k != "queries"} == {k: v for k, v in json.loads(text).items() if k != "queries"}
            assert [q["index"] for q in obj["queries"]] == [q["index"] for q in pr["queries"]]
            assert VA.Shape.of_proof(obj, 2).key() == VA.Shape.of_proof(pr, 2).key()
    good = PR.proof_to_json(proofs[0])
    bad = copy.deepcopy(proofs[0])
    bad["queries"][2]["trace"]["values"] = bad["queries"][2]["trace"]["values"][:-1]          # ragged
    assert VA.prepare_proof_text(json.dumps(bad)) is None
    bad = copy.deepcopy(proofs[0])
    bad["queries"][1]["fri"][0]["path"][1] = bad["queries"][1]["fri"][0]["path"][1][:3]
    assert VA.prepare_proof_text(json.dumps(bad)) is None
    for broken in (good.replace('"index":', '"index":-', 1), good.replace('"values":[', '"values":[1.5,', 1), good[:-2], good.replace('"queries"', '"q"'),
                   good.replace('"values":[', '"values":[18446744073709551616,', 1), "[1,2]", ""):
        assert VA.prepare_proof_text(broken) is None
    doc = b'{"a":[1,{"stark":2}],"stark":{"x":"}"} ,"z":1}'
    sb, se = native.json_key_span(doc, "stark")
    assert doc[sb:se] == b'{"x":"}"}'                        # the top-level member, not the nested one; braces inside strings do not count
    assert native.json_key_span(b'{"a":1}', "stark") is None
    # the engine: the two paths give the same prepared arrays (and therefore the same aggregation)
    monkeypatch.setattr(Engine, "FAST_PARSE_MIN", 0)
    _, whole_f, prep_f = Engine._parse_and_prepare(good)
    monkeypatch.setattr(Engine, "FAST_PARSE_MIN", 1 << 40)
    _, whole_s, prep_s = Engine._parse_and_prepare(good)
    assert all((a == b).all() for a, b in zip(prep_f["values"] + prep_f["paths"] + [prep_f["index"]], prep_s["values"] + prep_s["paths"] + [prep_s["index"]]))
    assert {k: v for k, v in whole_f.items() if k != "queries"} == {k: v for k, v in whole_s.items() if k != "queries"}
    shape = VA.Shape.of_proof(whole_f, 1)
    t1, p1 = VA.build_witness(shape, [whole_f], cpu, air.digest_words(), [prep_f])
    t2, p2 = VA.build_witness(shape, [whole_s], cpu, air.digest_words(), [prep_s])
    assert (t1 == t2).all() and (p1 == p2).all()


def test_publics_from_headers_is_what_the_witness_builder_returns(inner, aggregated, cpu):
    """GenFinalProof's link check (round-5 advisor item): the public inputs an honest aggregation STARK has are a function of the inner HEADERS
    alone -- VA.publics_from_headers reads no opening and gives the list build_witness returned beside the trace; one changed header word gives
    another list (or no list at all: a header that fails its own grinding)."""
    air, params, proofs = inner
    shape, _, _, _, pubs, agg = aggregated
    got = VA.publics_from_headers(shape, agg["inner"], air.digest_words(), cpu)
    assert got == [int(v) for v in pubs]
    for mutate in (lambda h: h["evals"]["z"][2].__setitem__(1, (h["evals"]["z"][2][1] + 1) % P),
                   lambda h: h["roots"]["trace"].__setitem__(0, (h["roots"]["trace"][0] + 1) % P),
                   lambda h: h["fri"]["final"][1].__setitem__(0, (h["fri"]["final"][1][0] + 1) % P),
                   lambda h: h["publics"].__setitem__(0, (h["publics"][0] + 1) % P)):
        bad = [strip_paths(h) for h in agg["inner"]]
        mutate(bad[1])
        try:
            assert VA.publics_from_headers(shape, bad, air.digest_words(), cpu) != got
        except ValueError:
            pass
    with pytest.raises(ValueError):
        VA.publics_from_headers(shape, agg["inner"][:1], air.digest_words(), cpu)


def test_product_header_verifier_agrees_with_the_checker(inner, aggregated, cpu, tables):
    """stark/verifier.py (what Engine.final() runs on the client's aggregated proof before wrapping it; constraint identity through the library's
    zp_program_eval_ext) against the checker's verifier in header-only mode: same verdict and same transcript outputs on a chunk proof (stage-2
    columns, grinding, inline public inputs) and on an aggregation STARK over two of them (a verifier AIR: ~10^2 sparse fixed columns, some with
    public-input entries, thousands of public inputs entering through their commitment); and it refuses what the queries do not cover."""
    from eigen_zeth_amd.stark import verifier as SV
    air, params, proofs = inner
    rc, mds = tables
    want = V.verify(proofs[0], air.program(), rc, mds, V.expectation(params.to_dict()), header_only=True)
    got = SV.verify_header(proofs[0], air, params, cpu)
    assert got["indices"] == want["indices"] and got["zeta"] == want["zeta"] and got["gamma"] == want["gamma"] and got["betas"] == want["betas"]
    shape, vair, ap, _, _, agg = aggregated
    outer = json.loads(PR.proof_to_json(agg["stark"]))
    NV = V.NV
    want = V.verify(outer, vair.program(), rc, mds, V.expectation(ap.to_dict()), header_only=True)
    got = SV.verify_header(outer, vair, ap, cpu)
    assert got["indices"] == want["indices"] and got["zeta"] == want["zeta"]
    # the library's evaluation of the statement at zeta = the checker's own interpreter over F_{p^3}, constraint by constraint
    from oracle.air_program import Program
    prog = Program(vair.program())
    Wt = vair.width + vair.width2
    zeta, N = want["zeta"], 1 << ap.logn
    wlast = pow(NV.root(ap.logn, outer["root32"]), N - 1, V.P)
    zh = [(v - (1 if i == 0 else 0)) % V.P for i, v in enumerate(NV.e3_pow(zeta, N))]
    ninv = pow(N, V.P - 2, V.P)
    l_first = NV.e3_mul([v * ninv % V.P for v in zh], NV.e3_inv([(zeta[0] - 1) % V.P, zeta[1], zeta[2]]))
    l_last = NV.e3_mul([v * ninv % V.P * wlast % V.P for v in zh], NV.e3_inv([(zeta[0] - wlast) % V.P, zeta[1], zeta[2]]))
    fixed_z = [l_first, l_last] + [prog.fixed_eval_ext(k, outer["publics"], zeta, ap.logn, outer["root32"]) for k in range(len(prog.fixed_cols))]
    cs = prog.evaluate_ext(outer["evals"]["z"][:Wt], outer["evals"]["zw"], fixed_z, list(outer["publics"]), [(zeta[0] - wlast) % V.P, zeta[1], zeta[2]])
    lib = native.program_eval_ext(vair.program(), outer["publics"], ap.logn, outer["root32"], zeta, outer["evals"]["z"][:Wt], outer["evals"]["zw"], threads=3)
    assert lib.tolist() == [[int(v) % V.P for v in c] for c in cs]
    # refused: an evaluation that is not the polynomial's, a final layer of too high a degree, another security level, a trimmed program
    for mutate, why in ((lambda p: p["evals"]["z"][1].__setitem__(0, (p["evals"]["z"][1][0] + 1) % V.P), "identity"),
                        (lambda p: p["evals"]["zw"][0].__setitem__(2, (p["evals"]["zw"][0][2] + 1) % V.P), "identity"),
                        (lambda p: p["fri"]["final"][0].__setitem__(3, (p["fri"]["final"][0][3] + 1) % V.P), "low degree"),
                        (lambda p: p["params"].__setitem__("n_queries", 1), "parameters"),
                        (lambda p: p["publics"].__setitem__(7, (p["publics"][7] + 1) % V.P), "identity"),
                        (lambda p: p["roots"].__setitem__("quotient", p["roots"]["quotient"][:3]), "root")):
        bad = copy.deepcopy(outer)
        mutate(bad)
        with pytest.raises(SV.Reject, match=why):
            SV.verify_header(bad, vair, ap, cpu)
        with pytest.raises((V.Reject, KeyError, IndexError, ValueError, TypeError)):
            V.verify(bad, vair.program(), rc, mds, V.expectation(ap.to_dict()), header_only=True)
    with pytest.raises(ValueError):
        native.program_eval_ext(vair.program()[:-2], outer["publics"], ap.logn, outer["root32"], zeta, outer["evals"]["z"][:Wt], outer["evals"]["zw"])
    with pytest.raises(ValueError):                                  # zeta on the trace domain
        native.program_eval_ext(vair.program(), outer["publics"], ap.logn, outer["root32"], [1, 0, 0], outer["evals"]["z"][:Wt], outer["evals"]["zw"])
