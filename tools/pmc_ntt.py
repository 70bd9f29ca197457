"""one NTT configuration under rocprofv3 --pmc (measurement tool)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn, cols = 24, 16
p = Prover(0)
for a in sys.argv[1:]:
    k, v = a.split("=")
    p.set_tuning(k, int(v))
x = np.random.default_rng(1).integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
d = p.upload(x); o = p.alloc(cols << logn)
for _ in range(3):
    p.ntt(d, o, logn, cols)
p.sync()
