import sys, time, json, cProfile, pstats
sys.path.insert(0, '/root/repo')
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory
from eigen_zeth_amd.stark import verifier as SV, verifier_air as VA
cfg = EngineConfig(air="chunk64", logn=20, chunks_per_block=1, groth16_seed="t")
eng = Engine(default_backend_factory(0), cfg)
ch = eng.gen_batch_chunks("w", [3, 4], 12345, "evm")
proofs = eng.gen_chunk_proofs("w", ch["task_id"], ch["chunk_count"], ch["batch_data"])
agg = eng.aggregate("w", proofs[0]["proof"], proofs[1]["proof"])
js, pub = eng.final("w", agg, "BN128", "1")
_, aggo, prep = eng._parse_and_prepare(agg)
outer = aggo["stark"]
shape = eng._own_shape(aggo)
air = VA.verifier_air(shape, *eng._tables(eng.be_bn128))
params = eng._agg_params(shape)
for _ in range(2):
    t = time.perf_counter(); SV.verify_header(outer, air, params, eng.be); print("verify_header ms", (time.perf_counter() - t) * 1e3)
pr = cProfile.Profile(); pr.enable(); SV.verify_header(outer, air, params, eng.be); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
print(json.dumps({k: round(v * 1e3, 1) for k, v in eng.stage_timings["final/w"].items()}))
