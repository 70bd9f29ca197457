"""where the host time of GenAggregatedProof / GenFinalProof goes (measurement tool): cProfile around one warm engine.aggregate and one
engine.final at the service's size.  usage: python tools/recursion_pyprofile.py [logn=20]"""
import cProfile, io, json, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = EngineConfig(air="chunk64", logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_rec_%d" % os.getuid()))
eng = Engine(default_backend_factory(0), cfg)
ch = eng.gen_batch_chunks("r", [1, 2], 12345, "evm")
proofs = eng.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])
for _ in range(3):
    agg = eng.aggregate("r", proofs[0]["proof"], proofs[1]["proof"])
    eng.final("r", agg, "BN128", "1")
for name, fn in (("aggregate", lambda: eng.aggregate("r", proofs[0]["proof"], proofs[1]["proof"])), ("final", lambda: eng.final("r", agg, "BN128", "1"))):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    fn()
    pr.disable()
    print("==== %s: %.1f ms under the profiler" % (name, (time.perf_counter() - t0) * 1e3))
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
