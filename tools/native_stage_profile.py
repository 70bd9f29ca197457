"""per-entry-point GPU time inside one zp_stark_prove call (HIP events of the library's own stage profiler; measurement tool)
usage: python tools/native_stage_profile.py [air=chunk64] [logn=20]"""
import json, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.native import Prover
name = sys.argv[1] if len(sys.argv) > 1 else "chunk64"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
air = AIR.get_air(name)
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42)
p = Prover(0)
d = p.upload(tr)
args = (air.name, air.program(), d, [int(v) for v in pub], logn, 1, 3, 5, 80, 20)
p.stark_prove(*args)
p.set_profiling(True)
t0 = time.perf_counter()
p.stark_prove(*args)
wall = time.perf_counter() - t0
st = p.stage_timings()
p.set_profiling(False)
acc = collections.OrderedDict()
for e in st if isinstance(st, list) else st.get("stages", []):
    acc.setdefault(e["stage"], [0, 0.0])
    acc[e["stage"]][0] += 1
    acc[e["stage"]][1] += e["ms"]
print(json.dumps({"air": name, "logn": logn, "wall_ms_trace_resident": round(wall * 1e3, 2),
                  "gpu_ms_by_entry_point": {k: [v[0], round(v[1], 3)] for k, v in acc.items()}}))
