"""LDE seam kernel variants (measurement tool): python tools/lde_ab2.py [logn] [cols]  -- alternates knob settings on one box"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
p = Prover(0)
N = 1 << logn
d = p.alloc(cols * N); o = p.alloc(2 * cols * N); o2 = p.alloc(2 * cols * N)
rng = np.random.default_rng(1)
for c0 in range(0, cols, 8):
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(min(8, cols - c0), N), dtype=np.uint64)
    p._chk(p.lib.zp_h2d(p.ctx, d.ptr + c0 * N * 8, x.ctypes.data, x.nbytes))
ref = None
for rep in range(3):
    for label, knobs in (("seam kernel", {"lde_seam": 1}), ("two launches", {"lde_seam": 0})):
        for k, v in knobs.items():
            p.set_tuning(k, v)
        p.lde(d, o, logn, 1, cols); p.sync()
        p.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(5):
            p.lde(d, o, logn, 1, cols)
        p.sync(); dt = (time.perf_counter() - t0) / 5
        by = {}
        for rl, ms in p.pass_timings():
            by.setdefault(rl, []).append(ms)
        p.set_profiling(False)
        print("%-16s LDE 2^%d x %d b=2: %7.3f ms  %6.0f GB/s algorithmic  per-launch ms %s" % (label, logn, cols, dt * 1e3, 24.0 * N * cols / dt / 1e9,
              {k: round(sum(v) / len(v), 3) for k, v in by.items()}), flush=True)
        if rep == 0:
            got = p.download(o, (2, 2 * N))
            if ref is None:
                ref = got
            else:
                print("   same words as the first form:", bool((got == ref).all()), flush=True)
p.set_tuning("lde_seam", 1)
