"""NTT time vs columns per launch (does a small working set let the Infinity Cache absorb the middle passes?)
measurement tool; run on the GPU box.  usage: python tools/ntt_chunk_sweep.py [logn] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = Prover(0)
N = 1 << logn
x = np.random.default_rng(1).integers(0, 2**63, size=(cols, N), dtype=np.uint64)
d = p.upload(x)
o = p.alloc(cols * N)
for chunk_log in (28, 27, 26, 25, 24, 23, 22):
    if chunk_log < logn:
        continue
    p.set_tuning("ntt_chunk_log", chunk_log)
    p.ntt(d, o, logn, cols); p.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        p.ntt(d, o, logn, cols)
    p.sync()
    dt = (time.perf_counter() - t0) / 5
    print("logn %d cols %d chunk 2^%d elems (%d cols/launch): %.3f ms  %.1f Gelem/s" % (logn, cols, chunk_log, 1 << (chunk_log - logn), dt * 1e3, cols * N / dt / 1e9), flush=True)
