"""one LDE configuration (blow-up 2, 2^24 rows x 8 columns = ONE launch per pass) under rocprofv3 --pmc (measurement tool): the six
pass kernels of zp_lde -- three of the inverse transform, three of the zero-padded forward transform -- with FETCH_SIZE / WRITE_SIZE
per launch, so that the traffic of the whole extension can be set against its 24 N algorithmic bytes per column"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn, cols = 24, 8
keep_coef = not (len(sys.argv) > 1 and sys.argv[1] == "nocoef")      # "nocoef": the caller does not want the coefficients -> the fused seam kernel
p = Prover(0)
x = np.random.default_rng(1).integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
d = p.upload(x); o = p.alloc(cols << (logn + 1)); c = p.alloc(cols << logn)
for _ in range(3):
    p.lde(d, o, logn, 1, cols, d_coef=c if keep_coef else None)
p.sync()
