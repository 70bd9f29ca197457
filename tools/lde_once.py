"""three LDE calls (2^24 rows, 32 columns, blow-up 2) for a kernel trace (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn, cols = 24, 32
p = Prover(0)
x = np.random.default_rng(1).integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
d = p.upload(x); o = p.alloc(cols << (logn + 1)); c = p.alloc(cols << logn)
for with_coef in (False, True):
    p.lde(d, o, logn, 1, cols, d_coef=c if with_coef else None); p.sync()
    t0 = time.perf_counter()
    for _ in range(3): p.lde(d, o, logn, 1, cols, d_coef=c if with_coef else None)
    p.sync()
    print("LDE 2^%d x %d b=2 %s: %.2f ms" % (logn, cols, "with d_coef" if with_coef else "no d_coef", (time.perf_counter() - t0) / 3 * 1e3), flush=True)
