import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
logm, W = 23, 32
a = p.upload(np.random.default_rng(1).integers(0, 2**63, size=(W, 1 << logm), dtype=np.uint64))
b = p.upload(np.random.default_rng(2).integers(0, 2**63, size=(3, 1 << logm), dtype=np.uint64))
o = p.alloc(3 << logm)
ev = np.random.default_rng(3).integers(0, 2**63, size=(W + 3, 3), dtype=np.uint64)
for zz in ([5, 6, 7], [123456789123, 987654321987, 555555555555]):
    p.deep_quotient(a, W, b, 3, logm, W, zz, [1, 2, 3], [4, 5, 6], ev, ev[:W], 49, o); p.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        p.deep_quotient(a, W, b, 3, logm, W, zz, [1, 2, 3], [4, 5, 6], ev, ev[:W], 49, o)
    p.sync()
    print("deep z=%s: %.2f ms" % (zz[:1], (time.perf_counter() - t0) / 3 * 1e3))
