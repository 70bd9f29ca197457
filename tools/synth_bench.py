"""Synthetic witness generation, host against device (measurement tool): python tools/synth_bench.py [logn] [chunks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
from eigen_zeth_amd.native import Prover
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W = 76
p = Prover(0)
t0 = time.perf_counter(); tr, pub = native.synth_trace(3, logn, W, 1); t1 = time.perf_counter()
print("host generator, one chunk 2^%d x %d: %.3f s (one thread)" % (logn, W, t1 - t0), flush=True)
for rep in range(2):
    t0 = time.perf_counter(); ck = p.synth_checkpoints(3, logn, W, list(range(1, n + 1))); t1 = time.perf_counter()
    d = None
    ts = []
    for i in range(n):
        a = time.perf_counter(); d, dpub = p.synth_trace_device(3, logn, W, i + 1, ckpt=ck, ckpt_index=i, out=d); p.sync(); ts.append(time.perf_counter() - a)
    print("device: checkpoints of %d chunks %.3f s; one trace from its checkpoints %.2f ms (median of %d)" % (n, t1 - t0, sorted(ts)[len(ts) // 2] * 1e3, n), flush=True)
    got = p.download(d, (W, 1 << logn)) if rep == 0 else None
    if rep == 0:
        ref, _ = native.synth_trace(3, logn, W, n)
        print("   last device trace == host trace:", bool((got == ref).all()), flush=True)
    ck.free()
