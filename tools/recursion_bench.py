"""GenAggregatedProof + GenFinalProof at the service's sizes on one GPU (measurement tool): two chunk64 proofs of 2^logn rows
at the default security -> aggregation STARK over the Merkle-verifier AIR -> final STARK (BN128-hash mode) over the
aggregated proof -> Groth16 wrap, with the engine's stage timings.   usage: python tools/recursion_bench.py [logn] [reps] [chunks: 2, or more = aggregate the whole batch]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory
from eigen_zeth_amd.stark import verifier_air as VA

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n_chunks = int(sys.argv[3]) if len(sys.argv) > 3 else 2      # > 2: the whole batch is aggregated (EngineConfig.aggregate_all_chunks)
cfg = EngineConfig(air="chunk64", logn=logn, groth16_logm=8, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_rec_%d" % os.getuid()),
                   aggregate_all_chunks=n_chunks > 2)
eng = Engine(default_backend_factory(0), cfg)
eng.groth16_keys()
ch = eng.gen_batch_chunks("r", list(range(1, n_chunks + 1)), 12345, "evm")
proofs = eng.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])
for rep in range(reps):
    t0 = time.perf_counter()
    agg = eng.aggregate("r", proofs[0]["proof"], proofs[-1]["proof"])
    t1 = time.perf_counter()
    fin, pub = eng.final("r", agg, "BN128", "479881985774944702531460751064278034642760119942")
    t2 = time.perf_counter()
    a = json.loads(agg)
    fs = json.loads(eng.final_starks["r"])
    print(json.dumps({"rep": rep, "inner_logn": logn, "inner_proofs_verified": len(a["inner"]), "aggregate_s": round(t1 - t0, 4), "final_s": round(t2 - t1, 4),
                      "aggregation_trace_logn": a["stark"]["params"]["logn"], "final_trace_logn": fs["params"]["logn"],
                      "aggregated_proof_bytes": len(agg), "final_stark_bytes": len(eng.final_starks["r"]),
                      "stages": {k: {kk: round(vv, 4) for kk, vv in v.items()} for k, v in eng.stage_timings.items() if k.startswith(("aggregate", "final"))}}),
          flush=True)
