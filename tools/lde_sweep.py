"""LDE timing with tuning variants (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn, cols = 24, 16
p = Prover(0)
x = np.random.default_rng(1).integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
d = p.upload(x); o = p.alloc(cols << (logn + 1))
def run(label, **tune):
    for k, v in tune.items(): p.set_tuning(k, v)
    p.lde(d, o, logn, 1, cols); p.sync()
    p.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(3): p.lde(d, o, logn, 1, cols)
    p.sync(); dt = (time.perf_counter() - t0) / 3
    by = {}
    for rl, ms in p.pass_timings(): by.setdefault(rl, []).append(ms)
    p.set_profiling(False)
    print("%-24s %7.2f ms  per-pass %s" % (label, dt * 1e3, {k: round(sum(v) / len(v), 3) for k, v in by.items()}), flush=True)
run("default")
run("L9 T=16", ntt_logt9=4)
run("L9 T=16 tpw=2", ntt_logt9=4, ntt_tpw=2)
