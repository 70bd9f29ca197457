"""zp_ood_eval (barycentric out-of-domain evaluation from resident values, csrc/stark.hip) in isolation: device time per entry-point call by the
library's stage clock, against the bytes it has to read.  Measurement tool.  usage: python tools/ood_bench.py [logn=22]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 22
p = Prover(0)
N = 1 << logn
z = [123456789, 987654321, 555]
for W, nxt in ((64, True), (12, True), (3, False), (16, True), (32, True)):
    d = p.alloc(W * N)
    p.lib.zp_dev_zero(p.ctx, d.ptr, W * N * 8)
    p.ood_eval(d, N, 1, W, logn, 1, z, nxt)          # weights built and cached
    p.set_profiling(True)
    p.stage_timings()
    ts = []
    for rep in range(5):
        p.sync(); t0 = time.perf_counter()
        p.ood_eval(d, N, 1, W, logn, 1, z, nxt)
        ts.append((time.perf_counter() - t0) * 1e3)
    st = p.stage_timings()
    p.set_profiling(False)
    dev = sorted(s["ms"] for s in st if s["stage"] == "ood_eval")
    gb = W * N * 8 / 1e9
    print("2^%d x %-2d next=%d: wall %.3f ms  device %.3f ms  columns %.2f GB -> %.0f GB/s" % (logn, W, nxt, sorted(ts)[2], dev[len(dev) // 2], gb, gb / (dev[len(dev) // 2] * 1e-3)), flush=True)
    d.free()
