"""LDE timing, one configuration (measurement tool; library variants through ZP_LIB_PATH): python tools/lde_ab.py [logn] [cols] [coef]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 32
coef = len(sys.argv) > 3 and sys.argv[3] == "coef"
p = Prover(0)
N = 1 << logn
d = p.alloc(cols * N); o = p.alloc(2 * cols * N); c = p.alloc(cols * N) if coef else None
rng = np.random.default_rng(1)
for c0 in range(0, cols, 8):
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(min(8, cols - c0), N), dtype=np.uint64)
    p._chk(p.lib.zp_h2d(p.ctx, d.ptr + c0 * N * 8, x.ctypes.data, x.nbytes))
for knob in ([("lde_seam", 1), ("lde_seam", 0)] if os.environ.get("ZP_LDE_SEAM_AB") else [None]):
    if knob:
        p.set_tuning(*knob)
    for rep in range(3):
        p.lde(d, o, logn, 1, cols, d_coef=c); p.sync()
        p.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(5):
            p.lde(d, o, logn, 1, cols, d_coef=c)
        p.sync(); dt = (time.perf_counter() - t0) / 5
        by = {}
        for rl, ms in p.pass_timings():
            by.setdefault(rl, []).append(ms)
        p.set_profiling(False)
        print("%s LDE 2^%d x %d b=2%s: %7.3f ms  %6.0f GB/s algorithmic  per-launch ms %s" % (("%s=%d" % knob) if knob else "", logn, cols, " +coef" if coef else "", dt * 1e3, 24.0 * N * cols / dt / 1e9,
              {k: round(sum(v) / len(v), 3) for k, v in by.items()}), flush=True)
