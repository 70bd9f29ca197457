"""Columns per launch (knob ntt_chunk_log: elements per ping-pong buffer) for NTT and LDE over sizes (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
rng = np.random.default_rng(1)
for logn, cols in ((20, 128), (21, 76), (22, 76), (23, 76), (24, 64), (25, 32)):
    N = 1 << logn
    d = p.alloc(cols * N); o = p.alloc(2 * cols * N)
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(1, N), dtype=np.uint64)
    for c in range(cols):
        p._chk(p.lib.zp_h2d(p.ctx, d.ptr + c * N * 8, x.ctypes.data, x.nbytes))
    for rep in range(2):
        for ck in (27, 28, 29):
            p.set_tuning("ntt_chunk_log", ck)
            out = []
            for what, fn in (("ntt", lambda: p.ntt(d, o, logn, cols)), ("lde", lambda: p.lde(d, o, logn, 1, cols))):
                fn(); p.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                p.sync(); out.append("%s %.3f ms" % (what, (time.perf_counter() - t0) / 5 * 1e3))
            print("2^%d x %d  chunk 2^%d  %s" % (logn, cols, ck, "  ".join(out)), flush=True)
    d.free(); o.free()
p.set_tuning("ntt_chunk_log", 0)
