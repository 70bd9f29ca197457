"""round-5 same-box A/Bs of the headline transform (2^24 rows x 64 columns, in place as bench.py runs it).  Measurement tool.
  1. columns per launch (knob ntt_chunk_log 24 .. 28): with ONE column per launch the two ping-pong buffers (128 MiB each) are rewritten by every
     column -- do they stay in the 256 MiB Infinity Cache and take passes 2 and 3 off HBM?
  2. the two-pass plan (ntt_maxl = 12: two radix-4096 passes, 1024-thread workgroups, 128-KiB tiles of 4 columns) against the default three passes
  3. the plain-copy ceiling: grid size and load/store policy of zp_hbm_copy_probe
usage: python tools/ntt_r5_ab.py > profiles/r5_ntt_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
logn, W = 24, 64
x = np.random.default_rng(3).integers(0, 2**63, size=(W, 1 << logn), dtype=np.uint64)
d = p.upload(x)
del x


def timed(reps=8):
    ts = []
    for r in range(reps + 2):
        p.sync()
        t0 = time.perf_counter()
        p.ntt(d, d, logn, W)
        p.sync()
        if r >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


print("# tools/ntt_r5_ab.py on one MI355X: forward NTT 2^24 x 64 in place, ms per step (median, min of 8)")
for rnd in range(1):
    for cl in (28, 27, 26, 25, 24):
        p.set_tuning("ntt_chunk_log", cl)
        med, mn = timed()
        print("three passes (8,8,8), %2d columns per launch (chunk 2^%d): %.3f ms  (min %.3f)  %.1f G elems/s" % ((1 << cl) >> logn, cl, med, mn, W * (1 << logn) / med / 1e6), flush=True)
p.set_tuning("ntt_chunk_log", 0)
for maxl, cl in ((12, 28), (12, 26), (9, 28)):
    p.set_tuning("ntt_maxl", maxl)
    p.set_tuning("ntt_chunk_log", cl)
    med, mn = timed()
    print("maxl 2^%d plan %s chunk 2^%d: %.3f ms  (min %.3f)  %.1f G elems/s" % (maxl, [q["radix_log"] for q in p.ntt_plan(logn)["passes"]], cl, med, mn, W * (1 << logn) / med / 1e6), flush=True)
p.set_tuning("ntt_maxl", 0)
p.set_tuning("ntt_chunk_log", 0)
src, dst = p.alloc(1 << 28), p.alloc(1 << 28)          # 2 GiB each
for nt in (1, 0):
    for grid, block, un in ((2048, 256, 4), (256, 256, 4), (256, 256, 8), (256, 512, 4), (256, 512, 8), (256, 1024, 4), (512, 256, 4), (512, 512, 4), (512, 256, 8),
                            (768, 256, 4), (1024, 256, 4), (1024, 256, 8), (4096, 256, 4)):
        p.set_tuning("copy_nt", nt)
        p.set_tuning("copy_grid", grid)
        p.set_tuning("copy_block", block)
        p.set_tuning("copy_unroll", un)
        ms = min(p.hbm_copy_probe(src, dst, 8 << 28, reps=5) for _ in range(3))
        print("plain copy 2 GiB -> 2 GiB, %s, %5d workgroups x %4d lanes x 16 B x %d in flight: %.3f ms = %.0f GB/s (read + write)"
              % ("non-temporal" if nt else "default policy", grid, block, un, ms, 2.0 * (8 << 28) / ms / 1e6), flush=True)
