"""device-time per C-ABI entry point (zp_stage_timings) for one STARK proof, next to the wall clock"""
import json, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR, prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "wide64"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
air = AIR.get_air(name)
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42)
be = HipBackend(0)
params = PR.StarkParams(logn, logb=1, fri_logf=3, fri_final_log=5, n_queries=80, pow_bits=20)   # the service default (100 bits)
PR.prove(air, tr, pub, params, be)
be.p.set_profiling(True)
tm = {}
t0 = time.perf_counter(); PR.prove(air, tr, pub, params, be, timings=tm); wall = time.perf_counter() - t0
rep = be.p.stage_timings(); be.p.pass_timings(); be.p.set_profiling(False)
agg = collections.OrderedDict()
for r in rep: agg[r["stage"]] = agg.get(r["stage"], 0.0) + r["ms"]
print([ (r["stage"], round(r["ms"],3)) for r in rep if r["stage"] in ("poly_eval_ext","lde","intt","deep_quotient")], file=sys.stderr)
print(json.dumps({"wall_ms": wall * 1e3, "device_ms_by_entry_point": {k: round(v, 3) for k, v in agg.items()},
                  "device_ms_total": round(sum(agg.values()), 3), "calls": len(rep),
                  "host_stage_ms": {k: round(v * 1e3, 2) for k, v in tm.items()}}))
