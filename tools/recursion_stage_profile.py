"""per-entry-point GPU time inside the two recursion STARKs (measurement tool): the aggregation STARK (Goldilocks mode, zp_stark_prove)
and the final STARK (BN128-hash mode, zp_stark_prove_bn128), both over the Merkle-verifier AIR, traces resident.
usage: python tools/recursion_stage_profile.py [inner logn=20]"""
import collections, json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory
from eigen_zeth_amd.stark import air as AIR, verifier_air as VA

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = EngineConfig(air="chunk64", logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_rec_%d" % os.getuid()))
eng = Engine(default_backend_factory(0), cfg)
ch = eng.gen_batch_chunks("r", [1, 2], 12345, "evm")
proofs = [json.loads(p["proof"]) for p in eng.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])]


def profile(be, inner, params_of, label, inner_air):
    rc, mds = eng._tables(be)
    shape = VA.Shape.of_proof(inner[0], len(inner))
    vair = VA.verifier_air(shape, rc, mds)
    trace, pubs = VA.build_witness(shape, inner, be, inner_air.digest_words())
    params = params_of(shape)
    d = trace if hasattr(trace, "ptr") else be.p.upload(trace)      # the GPU backend assembles the trace in HBM
    fn = be.p.stark_prove_bn128 if params.hash == "bn128" else be.p.stark_prove
    args = (vair.name, vair.program(), d, [int(v) for v in pubs], params.logn, params.logb, params.fri_logf, params.fri_final_log, params.n_queries)
    if params.hash != "bn128":
        args += (params.pow_bits,)
    text = fn(*args)
    be.p.set_profiling(True)
    t0 = time.perf_counter()
    text = fn(*args)
    wall = time.perf_counter() - t0
    st = be.p.stage_timings()
    be.p.set_profiling(False)
    acc = collections.OrderedDict()
    for e in st if isinstance(st, list) else st.get("stages", []):
        acc.setdefault(e["stage"], [0, 0.0])
        acc[e["stage"]][0] += 1
        acc[e["stage"]][1] += e["ms"]
    print(json.dumps({"stark": label, "trace_logn": params.logn, "wall_ms_trace_resident": round(wall * 1e3, 2),
                      "gpu_ms_by_entry_point": {k: [v[0], round(v[1], 3)] for k, v in acc.items()}}), flush=True)
    return json.loads(text)


agg = profile(eng.be, proofs, lambda sh: VA.aggregation_params(sh, cfg.agg_queries, cfg.fri_logf, cfg.fri_final_log, cfg.agg_pow_bits), "aggregation (Goldilocks mode)",
              AIR.get_air("chunk64"))
profile(eng.be_bn128, [agg], lambda sh: eng.final_stark_params(agg), "final (BN128-hash mode)",
        VA.verifier_air(VA.Shape.of_proof(proofs[0], len(proofs)), *eng._tables(eng.be)))
