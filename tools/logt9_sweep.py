"""Radix-512 passes with 32-wide (128 KiB) against 16-wide (64 KiB) tiles: NTT / iNTT / LDE over sizes whose plans hold a 9-digit (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
rng = np.random.default_rng(1)
for logn, cols in ((17, 256), (18, 128), (25, 16), (26, 8), (27, 4), (21, 64)):
    N = 1 << logn
    d = p.alloc(cols * N); o = p.alloc(2 * cols * N)
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(1, N), dtype=np.uint64)
    for c in range(cols):
        p._chk(p.lib.zp_h2d(p.ctx, d.ptr + c * N * 8, x.ctypes.data, x.nbytes))
    for rep in range(2):
        for label, knobs in (("L9 T=32", {"ntt_logt9": 5}), ("L9 T=16", {"ntt_logt9": 4})):
            for k, v in knobs.items():
                p.set_tuning(k, v)
            out = []
            for what, fn in (("ntt", lambda: p.ntt(d, o, logn, cols)), ("intt", lambda: p.intt(d, o, logn, cols)), ("lde", lambda: p.lde(d, o, logn, 1, cols))):
                fn(); p.sync()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                p.sync(); out.append("%s %.3f ms" % (what, (time.perf_counter() - t0) / 5 * 1e3))
            print("2^%d x %d  %-8s %s   plan %s" % (logn, cols, label, "  ".join(out), [q["radix_log"] for q in p.ntt_plan(logn)["passes"]]), flush=True)
    d.free(); o.free()
