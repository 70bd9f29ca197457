"""Timing sweep of NTT kernel variants on one GPU (measurement tool; run on the GPU box).
usage: python tools/ntt_sweep.py [logn] [cols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 16
p = Prover(0)
N = 1 << logn
rng = np.random.default_rng(1)
x = rng.integers(0, 2**63, size=(cols, N), dtype=np.uint64)
d = p.upload(x)
o = p.alloc(cols * N)

def run(label, reps=5, **tune):
    for k, v in tune.items():
        p.set_tuning(k, v)
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
    gbs = 16.0 * N * cols / dt / 1e9
    print("%-34s %8.3f ms/transform  %7.1f Gelem/s  alg %6.0f GB/s  per-pass ms %s" % (label, dt * 1e3, cols * N / dt / 1e9, gbs, per), flush=True)
    for k in tune:
        p.set_tuning(k, {"ntt_logt": 4, "ntt_tpw": 2, "ntt_tw1": 26}.get(k, 0))

run("warm-up")
for rep in range(3):        # alternating, three times: the first measurements of a process run on a cold clock
    run("T=16 tpw=2 (default), first-pass table", ntt_tpw=2)
    run("T=16 tpw=1", ntt_tpw=1)
    run("T=16 tpw=4", ntt_tpw=4)
    run("T=32 tpw=2", ntt_logt=5)
    run("first pass: per-lane chains", ntt_tw1=0)
