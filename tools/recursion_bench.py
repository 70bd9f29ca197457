"""wall-clock of the two recursion requests at the service's size (measurement tool): GenAggregatedProof over two 2^logn-row chunk proofs
and GenFinalProof over the result, through service/engine.py on one GPU, stage by stage; the sizes of what they return.
usage: python tools/recursion_bench.py [logn=20] [reps=4]"""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = EngineConfig(air="chunk64", logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_rec_%d" % os.getuid()))
eng = Engine(default_backend_factory(0), cfg)
ch = eng.gen_batch_chunks("r", [1, 2], 12345, "evm")
proofs = eng.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])
for rep in range(reps):
    t0 = time.perf_counter()
    agg = eng.aggregate("r", proofs[0]["proof"], proofs[1]["proof"])
    t1 = time.perf_counter()
    fin, pub = eng.final("r", agg, "BN128", "479881985774944702531460751064278034642760119942")
    t2 = time.perf_counter()
    a = json.loads(agg)
    print(json.dumps({"rep": rep, "inner_logn": logn, "aggregate_ms": round((t1 - t0) * 1e3, 1), "final_ms": round((t2 - t1) * 1e3, 1),
                      "aggregate_stages_ms": {k: round(v * 1e3, 1) for k, v in eng.stage_timings["aggregate/r"].items()},
                      "final_stages_ms": {k: round(v * 1e3, 1) for k, v in eng.stage_timings["final/r"].items()},
                      "chunk_proof_bytes": len(proofs[0]["proof"]), "aggregated_proof_bytes": len(agg),
                      "aggregated_proof_inner_bytes": len(json.dumps(a["inner"], separators=(",", ":"))),
                      "aggregation_stark_bytes": len(json.dumps(a["stark"], separators=(",", ":"))), "agg_trace_logn": a["stark"]["params"]["logn"],
                      "agg_publics": len(a["stark"]["publics"]), "final_stark_bytes": len(eng.final_starks["r"]),
                      "final_trace_logn": json.loads(eng.final_starks["r"])["params"]["logn"]}), flush=True)
