"""A/B of the NTT passes' arithmetic forms on one GPU (measurement tool): limb form (gl_limb.hpp, knob ntt_limb = 1) against the canonical
carry-chain butterflies (ntt_limb = 0), alternating, same box, same buffers; checks that both give the same words.
usage: python tools/ntt_limb_ab.py [logn] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p = Prover(0)
N = 1 << logn
rng = np.random.default_rng(1)
d = p.alloc(cols * N)
o = p.alloc(cols * N)
for c0 in range(0, cols, 8):
    x = rng.integers(0, 0xFFFFFFFF00000001, size=(min(8, cols - c0), N), dtype=np.uint64)
    p._chk(p.lib.zp_h2d(p.ctx, d.ptr + c0 * N * 8, x.ctypes.data, x.nbytes))


def run(label, limb, reps=10):
    p.set_tuning("ntt_limb", limb)
    p.ntt(d, o, logn, cols); p.sync()
    p.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        p.ntt(d, o, logn, cols)
    p.sync()
    dt = (time.perf_counter() - t0) / reps
    pt = p.pass_timings()
    p.set_profiling(False)
    by = {}
    for rl, ms in pt:
        by.setdefault(rl, []).append(ms)
    per = {k: round(sum(v) / len(v), 3) for k, v in by.items()}
    print("%-28s %8.3f ms/transform  %7.1f Gelem/s  alg %6.0f GB/s  per-launch ms %s" % (label, dt * 1e3, cols * N / dt / 1e9, 16.0 * N * cols / dt / 1e9, per), flush=True)
    return p.download(o, (cols, N))[:2].copy()


run("warm-up", 1, 2)
for rep in range(3):
    a = run("limb form (ntt_limb=1)", 1)
    b = run("canonical (ntt_limb=0)", 0)
    print("   same words:", bool((a == b).all()), flush=True)
# inverse transform and an extension, once each way
for limb in (1, 0):
    p.set_tuning("ntt_limb", limb)
    t0 = time.perf_counter(); p.intt(d, o, logn, min(cols, 32)); p.sync(); t1 = time.perf_counter()
    print("intt x%d limb=%d: %.3f ms" % (min(cols, 32), limb, (t1 - t0) * 1e3), flush=True)
