"""The STARK-verifier AIR on the MI355X: witness kernel (zp_poseidon_trace), the aggregation STARK through the GPU backend and
through zp_stark_prove (periodic fixed columns read as one extended period: zp_fixed_columns), byte-identical to the CPU
checker's proof and accepted by the independent verifiers; then GenAggregatedProof / GenFinalProof through the engine at the
service's parameters (proto/prover/v1/prover.proto:115-148; client src/prover/provider.rs:422-503)."""
import copy
import json

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from eigen_zeth_amd.stark import verifier_air as VA
from eigen_zeth_amd.stark.backend_hip import HipBackend
from oracle import aggregate_verify as AV
from oracle import oracle as O
from oracle import stark_verify as V
from oracle.stark_cpu import CpuBackend

pytestmark = pytest.mark.gpu
P = O.P


def strip_paths(proof):
    """an inner proof as an aggregated proof carries it: the header, no openings"""
    return {k: copy.deepcopy(v) for k, v in proof.items() if k != "queries"}


@pytest.fixture(scope="module")
def hip(prover):
    return HipBackend(prover=prover)


@pytest.mark.parametrize("count", [1, 5, 128, 129, 1000])
def test_poseidon_trace_kernel_matches_oracle(prover, tables, count):
    rc, mds = tables
    x = O.random_field((count, 12), 900 + count)
    x[0, :3] = np.array([0, P - 1, 1], dtype=np.uint64)
    d_in, d_out = prover.upload(x), prover.alloc(24 * 32 * count)
    prover.poseidon_trace(d_in, count, d_out, d_out.offset(12 * 32 * count), 32 * count)
    got = prover.download(d_out, (24, 32 * count))
    st, cu = O.poseidon_trace(x, rc, mds)
    assert (got[:12] == st).all() and (got[12:] == cu).all()


def test_poseidon_trace_kernel_injected_tables(prover, tables):
    rc, mds = tables
    rc2 = O.random_field((360,), 29)
    mds2 = (O.random_field((144,), 30) % np.uint64(1 << 20)).astype(np.uint64)
    x = O.random_field((70, 12), 31)
    try:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc2)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds2)
        d_in, d_out = prover.upload(x), prover.alloc(24 * 32 * 70)
        prover.poseidon_trace(d_in, 70, d_out, d_out.offset(12 * 32 * 70), 32 * 70)
        got = prover.download(d_out, (24, 32 * 70))
        st, cu = O.poseidon_trace(x, rc2, mds2)
        assert (got[:12] == st).all() and (got[12:] == cu).all()
    finally:
        prover.set_constants(native.ZP_CONST_POSEIDON_RC, rc)
        prover.set_constants(native.ZP_CONST_POSEIDON_MDS, mds)


@pytest.mark.parametrize("airname,logn,nq", [("chunk16", 6, 4), ("wide8", 9, 6)])
def test_aggregation_stark_gpu_equals_cpu_and_verifies(hip, tables, airname, logn, nq):
    rc, mds = tables
    cpu = CpuBackend(rc, mds)
    air = AIR.get_air(airname)
    params = PR.StarkParams(logn, 1, 2, 3, nq, pow_bits=4)
    proofs = []
    for seed in (3, 4):
        tr, pub = native.synth_trace(air.trace_kind, logn, air.width, seed)
        proofs.append(json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, hip))))
    shape = VA.Shape.of_proof(proofs[0], 2)
    vair = VA.verifier_air(shape, rc, mds)
    d_gpu, pubs = VA.build_witness(shape, proofs, hip, air.digest_words())               # assembled in HBM (zp_poseidon_trace, native arithmetic columns)
    t_gpu = hip.p.download(d_gpu, d_gpu.shape)                                           # ... through ONE library call: zp_recursion_witness
    t_cpu, pubs_c = VA.build_witness(shape, proofs, cpu, air.digest_words())
    assert (t_gpu == t_cpu).all() and (pubs == pubs_c).all()
    d_step, pubs_s = VA.build_witness(shape, proofs, hip, air.digest_words(), keep={})   # the step-by-step path over the same GPU backend
    assert (hip.p.download(d_step, d_step.shape) == t_cpu).all() and (pubs_s == pubs_c).all()
    d_step.free()
    for mutate, what in ((lambda p: p[1]["queries"][1]["fri"][0]["values"].__setitem__(2, p[1]["queries"][1]["fri"][0]["values"][2] ^ 1), "hash"),
                         (lambda p: p[0]["queries"][2]["trace"]["path"][1].__setitem__(0, p[0]["queries"][2]["trace"]["path"][1][0] ^ 1), "hash"),
                         (lambda p: p[0]["queries"][0].__setitem__("index", p[0]["queries"][0]["index"] ^ 1), "transcript"),
                         (lambda p: p[1]["evals"]["z"][1].__setitem__(0, p[1]["evals"]["z"][1][0] ^ 1), "transcript")):
        bad = copy.deepcopy(proofs)
        mutate(bad)
        with pytest.raises(ValueError, match="no accepting witness"):                     # refused by the library, whatever is wrong
            VA.build_witness(shape, bad, hip, air.digest_words())
    ap = VA.aggregation_params(shape, n_queries=5, fri_final_log=3)
    p_cpu = PR.proof_to_json(PR.prove(vair, t_cpu, pubs, ap, cpu))
    p_gpu = PR.proof_to_json(PR.prove(vair, t_gpu, pubs, ap, hip))
    assert p_gpu == p_cpu
    assert hip.prove_native(vair, d_gpu, pubs, ap) == p_cpu                    # zp_stark_prove on the device-resident trace: the same bytes
    if airname == "chunk16":
        # round 5: the verifier AIR through a GENERATED constraint kernel (sparse periodic fixed columns read as one extended period each):
        # compiled on request (the service's prewarm does it for its two recursion programs), then the one-call prover gives the same bytes
        from eigen_zeth_amd.stark.backend_hip import HipBackend
        hk = HipBackend(0)
        assert hk.compile_air_kernel(vair)
        assert hk.prove_native(vair, t_cpu, pubs, ap) == p_cpu
        t = hk.p.stage_timings() if hasattr(hk.p, "stage_timings") else None
    agg = {"kind": "aggregated", "inner": [strip_paths(p) for p in proofs], "stark": json.loads(p_gpu)}
    assert AV.verify(agg, air.program(), vair.program(), rc, mds, V.expectation(params.to_dict()), V.expectation(ap.to_dict()), shape.n_slots())
    t_bad = t_gpu.copy()
    t_bad[VA.S0 + 1, 32 * 3 + 7] = (int(t_bad[VA.S0 + 1, 32 * 3 + 7]) + 1) % P
    with pytest.raises(V.Reject):
        V.verify(json.loads(hip.prove_native(vair, t_bad, pubs, ap)), vair.program(), rc, mds, V.expectation(ap.to_dict()))


def test_engine_aggregate_and_final_prove_what_they_name(tables, tmp_path):
    """the service's path at its default security (80 queries + 20 bits inner, 50 queries x blow-up 4 for both recursion
    layers), chunk AIR with 64 + 12 columns: the aggregated proof passes the checker's aggregate verifier, a tampered chunk
    proof is refused (application error, no proof), the final STARK (BN128-hash mode) verifies under the verifier AIR of the
    aggregated proof's shape with exactly its roots and indices as public inputs"""
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    rc, mds = tables
    cfg = EngineConfig(air="chunk64", logn=14, chunks_per_block=1, crs_dir=str(tmp_path / "crs"))
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("agg", [11, 12, 13], 12345, "evm")
    proofs = eng.gen_chunk_proofs("agg", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    text = eng.aggregate("agg", proofs[0]["proof"], proofs[-1]["proof"])
    agg = json.loads(text)
    assert agg["kind"] == "aggregated" and "standin" not in text and len(agg["inner"]) == 2
    sh = VA.Shape.from_dict(agg["shape"])
    vair = VA.verifier_air(sh, rc, mds)
    assert vair.digest() == agg["verifier_air_digest"]
    inner_exp = V.expectation(eng.stark_params(14).to_dict())
    outer_exp = V.expectation(VA.aggregation_params(sh, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log).to_dict())
    assert outer_exp["n_queries"] * outer_exp["logb"] >= 100
    assert AV.verify(agg, AIR.get_air("chunk64").program(), vair.program(), rc, mds, inner_exp, outer_exp, sh.n_slots())
    bad = json.loads(proofs[0]["proof"])
    bad["queries"][7]["trace"]["values"][3] ^= 1
    with pytest.raises(ValueError, match="no accepting witness"):
        eng.aggregate("agg2", json.dumps(bad), proofs[-1]["proof"])
    # one chunk: the client sends the same proof twice -> verified once
    one = json.loads(eng.aggregate("agg3", proofs[1]["proof"], proofs[1]["proof"]))
    assert len(one["inner"]) == 1 and one["shape"]["n_proofs"] == 1
    final, pub = eng.final("agg", text, "BN128", "479881985774944702531460751064278034642760119942")
    fsp = json.loads(eng.final_starks["agg"])
    fsh = VA.Shape.of_proof(agg["stark"], 1)
    fair = VA.verifier_air(fsh, rc, mds)
    assert fsp["air_digest"] == fair.digest() and [int(v) for v in fsp["publics"]][:fsh.merkle_pubs()] == VA.expected_publics(fsh, [agg["stark"]])
    assert V.verify(fsp, fair.program(), rc, mds, V.expectation(eng.final_stark_params(agg["stark"]).to_dict()), bn254_poseidon_params(17))
    # the final layer as a recursion layer: the aggregated proof's STARK without its paths + the final STARK = that STARK verifies
    assert all("queries" not in h for h in agg["inner"])           # succinct: the aggregated proof carries headers + ONE STARK
    hdr = {k: v for k, v in agg["stark"].items() if k != "queries"}
    assert AV.verify({"inner": [hdr], "stark": fsp}, vair.program(), fair.program(), rc, mds, outer_exp,
                     V.expectation(eng.final_stark_params(agg["stark"]).to_dict()), fsh.n_slots(), bn254_poseidon_params(17))
    with pytest.raises(ValueError):
        eng.final("x", json.dumps({"kind": "something-else"}), "BN128", "1")
    # the same request with the final STARK as ONE proof over two / eight ranks of the process (EngineConfig.final_ranks: zp_stark_prove_sharded_bn128
    # on an in-process communicator; the verifier AIR's 47 columns do not divide): the same final STARK, a wrap from rank 0's openings record
    # (the same engine with the knob turned: a second engine would make the wrap key again -- seconds at stage B-2's 3.2 M constraints)
    want_fs = eng.final_starks["agg"]
    for ranks in (2, 8):
        assert EngineConfig(air="chunk64", logn=14, final_ranks=ranks).final_ranks == ranks
        eng.cfg.final_ranks = ranks
        try:
            final_r, pub_r = eng.final("agg", text, "BN128", "479881985774944702531460751064278034642760119942")
        finally:
            eng.cfg.final_ranks = 1
        assert eng.final_starks["agg"] == want_fs and pub_r == pub
    with pytest.raises(ValueError):
        EngineConfig(air="chunk64", logn=14, final_ranks=3)
    print("stage timings:", json.dumps({k: v for k, v in eng.stage_timings.items() if k.startswith(("aggregate", "final"))}))


def test_engine_can_aggregate_every_chunk_of_a_batch(tables, tmp_path):
    """aggregate_all_chunks: the request still names the first and the last chunk proof (src/prover/provider.rs:385-388), the
    aggregation STARK verifies all of them (n_proofs = chunk count) and the final STARK sits on top as usual"""
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    rc, mds = tables
    cfg = EngineConfig(air="chunk16", logn=10, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), n_queries=12, pow_bits=4,
                       agg_queries=6, final_queries=4, aggregate_all_chunks=True)
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("all", [21, 22, 23, 24, 25], 12345, "evm")
    proofs = eng.gen_chunk_proofs("all", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    assert len(proofs) == 5
    text = eng.aggregate("all", proofs[0]["proof"], proofs[-1]["proof"])
    agg = json.loads(text)
    assert len(agg["inner"]) == 5 and agg["shape"]["n_proofs"] == 5
    assert [h["chunk"] for h in agg["inner"]] == [json.loads(p["proof"])["chunk"] for p in proofs]
    sh = VA.Shape.from_dict(agg["shape"])
    vair = VA.verifier_air(sh, rc, mds)
    inner_exp = V.expectation(eng.stark_params(10).to_dict())
    outer_exp = V.expectation(VA.aggregation_params(sh, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log).to_dict())
    assert AV.verify(agg, AIR.get_air("chunk16").program(), vair.program(), rc, mds, inner_exp, outer_exp, sh.n_slots())
    # two proofs that are not the ends of a known batch: exactly those two
    two = json.loads(eng.aggregate("all", proofs[1]["proof"], proofs[3]["proof"]))
    assert len(two["inner"]) == 2
    final, pub = eng.final("all", text, "BN128", "479881985774944702531460751064278034642760119942")
    fsp = json.loads(eng.final_starks["all"])
    fsh = VA.Shape.of_proof(agg["stark"], 1)
    assert V.verify(fsp, VA.verifier_air(fsh, rc, mds).program(), rc, mds, V.expectation(eng.final_stark_params(agg["stark"]).to_dict()),
                    bn254_poseidon_params(17))


def test_engine_folds_aggregated_proofs_again(tables, tmp_path):
    """GenAggregatedProof on two AGGREGATED proofs (level 2): the service's output passes the checker's tree verifier down to the
    four chunk proofs, GenFinalProof sits on top of it as on any aggregated proof, and mixed inputs are an application error"""
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    rc, mds = tables
    cfg = EngineConfig(air="chunk16", logn=10, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), n_queries=12, pow_bits=4,
                       agg_queries=6, final_queries=4)
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("t", [31, 32, 33, 34], 12345, "evm")
    proofs = eng.gen_chunk_proofs("t", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    ab = eng.aggregate("t", proofs[0]["proof"], proofs[1]["proof"])
    cd = eng.aggregate("t", proofs[2]["proof"], proofs[3]["proof"])
    top_text = eng.aggregate("t", ab, cd)
    top = json.loads(top_text)
    assert top["level"] == 2 and len(top["children"]) == 2 and [len(c["inner"]) for c in top["children"]] == [2, 2]
    assert top["inner"][0]["air_digest"] == json.loads(ab)["verifier_air_digest"]

    def program_of_shape(d):
        sh = VA.Shape.from_dict(d)
        return VA.verifier_air(sh, rc, mds).program(), sh.n_slots()

    def expect_of_shape(d):
        return V.expectation(VA.aggregation_params(VA.Shape.from_dict(d), cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log).to_dict())
    args = (AIR.get_air("chunk16").program(), V.expectation(eng.stark_params(10).to_dict()), rc, mds, program_of_shape, expect_of_shape)
    assert AV.verify_tree(top, *args)
    assert AV.verify_tree(json.loads(ab), *args)
    with pytest.raises(ValueError, match="cannot be folded"):
        eng.aggregate("t", ab, proofs[2]["proof"])
    final, pub = eng.final("t", top_text, "BN128", "479881985774944702531460751064278034642760119942")
    fsp = json.loads(eng.final_starks["t"])
    fsh = VA.Shape.of_proof(top["stark"], 1)
    assert V.verify(fsp, VA.verifier_air(fsh, rc, mds).program(), rc, mds, V.expectation(eng.final_stark_params(top["stark"]).to_dict()),
                    bn254_poseidon_params(17))
