"""plan digit order (larger radix first / last): NTT at sizes with unequal digits and the LDE (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
rng = np.random.default_rng(1)
def t(fn, reps=3):
    fn(); p.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    p.sync(); return (time.perf_counter() - t0) / reps * 1e3
for logn, cols in ((21, 64), (22, 64), (23, 32), (25, 16)):
    x = rng.integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
    d = p.upload(x); o = p.alloc(cols << logn)
    ref = None
    for order in (1, 2):
        p.set_tuning("ntt_order", order)
        ms = t(lambda: p.ntt(d, o, logn, cols))
        y = p.download(o, (cols, 1 << logn))[:, ::4097].copy()
        ref = y if ref is None else ref
        print("NTT 2^%d x %d order %d: %.3f ms  same=%s" % (logn, cols, order, ms, bool((y == ref).all())), flush=True)
    d.free(); o.free()
for logn, cols in ((20, 64), (24, 32)):
    x = rng.integers(0, 2**63, size=(cols, 1 << logn), dtype=np.uint64)
    d = p.upload(x); o = p.alloc(cols << (logn + 1))
    ref = None
    for order in (1, 2):
        p.set_tuning("ntt_order", order)
        ms = t(lambda: p.lde(d, o, logn, 1, cols))
        y = p.download(o, (cols, 2 << logn))[:, ::4097].copy()
        ref = y if ref is None else ref
        print("LDE 2^%d x %d b=2 order %d: %.3f ms  same=%s" % (logn, cols, order, ms, bool((y == ref).all())), flush=True)
    d.free(); o.free()
