"""One-call prover with the AIR's constraints through the interpreter against the generated kernel (zp_stark_set_air_kernel), same proof text
(measurement tool): python tools/air_kernel_ab.py [air=chunk64] [logn=20]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark.backend_hip import HipBackend
name = sys.argv[1] if len(sys.argv) > 1 else "chunk64"
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
air = AIR.get_air(name)
tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 42)
be = HipBackend(0)
p = be.p
d = p.upload(tr)
args = (air.name, air.program(), d, [int(v) for v in pub], logn, 1, 3, 5, 80, 20)
texts = {}
for rep in range(3):
    for label, fn in (("interpreter", None), ("generated kernel", be._airlib(air))):
        p.set_air_kernel(air.program(), fn)
        p.stark_prove(*args)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); texts[label] = p.stark_prove(*args); ts.append(time.perf_counter() - t0)
        print("%-17s zp_stark_prove %s x 2^%d: median %.2f ms  min %.2f ms" % (label, name, logn, sorted(ts)[2] * 1e3, min(ts) * 1e3), flush=True)
print("same proof text:", texts["interpreter"] == texts["generated kernel"])
