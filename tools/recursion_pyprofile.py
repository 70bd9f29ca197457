"""where the host time of GenAggregatedProof / GenFinalProof goes (measurement tool): cProfile around engine.aggregate / engine.final
after a warm-up round.   usage: python tools/recursion_pyprofile.py [logn]"""
import cProfile, io, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = EngineConfig(air="chunk64", logn=logn, groth16_logm=8, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_rec_%d" % os.getuid()))
eng = Engine(default_backend_factory(0), cfg)
eng.groth16_keys()
ch = eng.gen_batch_chunks("r", [1, 2], 12345, "evm")
proofs = eng.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])
addr = "479881985774944702531460751064278034642760119942"
for _ in range(2):
    agg = eng.aggregate("r", proofs[0]["proof"], proofs[-1]["proof"])
    eng.final("r", agg, "BN128", addr)
for name, fn in (("aggregate", lambda: eng.aggregate("r", proofs[0]["proof"], proofs[-1]["proof"])),
                 ("final", lambda: eng.final("r", agg, "BN128", addr))):
    t0 = time.perf_counter()
    fn()
    wall = time.perf_counter() - t0
    pr = cProfile.Profile()
    pr.enable()
    fn()
    pr.disable()
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
    print("==== %s: %.1f ms without the profiler" % (name, wall * 1e3))
    print("\n".join(l[:150] for l in out.getvalue().splitlines()[4:40]), flush=True)
