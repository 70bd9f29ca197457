"""GenFinalProof at the service's parameters, three times, for the profiler (measurement tool):
    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -- python3 tools/final_profile.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 14
cfg = EngineConfig(air="chunk64", logn=logn, chunks_per_block=1, groth16_seed="profile")
eng = Engine(default_backend_factory(0), cfg)
ch = eng.gen_batch_chunks("w", [3, 4], 12345, "evm")
proofs = eng.gen_chunk_proofs("w", ch["task_id"], ch["chunk_count"], ch["batch_data"])
agg = eng.aggregate("w", proofs[0]["proof"], proofs[1]["proof"])
for rep in range(4):
    t0 = time.perf_counter()
    eng.final("w", agg, "BN128", str(1000 + rep))
    print(json.dumps({"rep": rep, "final_ms": round((time.perf_counter() - t0) * 1e3, 1),
                      "stages_ms": {k: round(v * 1e3, 2) for k, v in eng.stage_timings["final/w"].items()}}), flush=True)
# the five MSMs of the wrap one after the other instead of on five streams (knob g16_parallel)
eng.be.p.set_tuning("g16_parallel", 0)
for rep in range(3):
    t0 = time.perf_counter()
    eng.final("w", agg, "BN128", str(2000 + rep))
    print(json.dumps({"g16_parallel": 0, "rep": rep, "final_ms": round((time.perf_counter() - t0) * 1e3, 1),
                      "stages_ms": {k: round(v * 1e3, 2) for k, v in eng.stage_timings["final/w"].items() if k.startswith("groth16")}}), flush=True)
eng.be.p.set_tuning("g16_parallel", 1)
