"""round 6, review item 4: the two-pass plan [12, 12] with TWO workgroups per CU (knob ntt_logt12 = 1: 512-thread workgroups, 64-KiB tiles of
2 columns, 16-byte runs) against the one-workgroup form (ntt_logt12 = 2: 1024 threads, 128-KiB tiles of 4 columns, 32-byte runs) and the
default three passes; forward NTT 2^24 x 64 in place, same box.  Every plan's output is compared with the default plan's first.
usage: python tools/ntt_two_pass_2wg_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
logn, W = 24, 64
x = np.random.default_rng(3).integers(0, 2**63, size=(4, 1 << logn), dtype=np.uint64) % np.uint64(0xFFFFFFFF00000001)
ref = None
for maxl, lt in ((0, 2), (12, 2), (12, 1)):
    p.set_tuning("ntt_maxl", maxl)
    p.set_tuning("ntt_logt12", lt)
    d = p.upload(x)
    p.ntt(d, d, logn, 4)
    y = p.download(d, x.shape)
    d.free()
    if ref is None:
        ref = y
    assert os.environ.get("ZP_AB_NOCHECK") or (y == ref).all(), "plan maxl=%d logt12=%d differs from the default plan" % (maxl, lt)
print("# all three plans write the same words (4 columns of 2^24)")
d = p.upload(np.random.default_rng(4).integers(0, 2**63, size=(W, 1 << logn), dtype=np.uint64))


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


for rnd in range(2):
    for maxl, lt, name in ((0, 2, "three passes (8,8,8), 256-thread workgroups, four per CU"),
                           (12, 2, "two passes (12,12), ONE 1024-thread workgroup per CU, 128-KiB tiles, 32-byte runs"),
                           (12, 1, "two passes (12,12), TWO 512-thread workgroups per CU, 64-KiB tiles, 16-byte runs")):
        p.set_tuning("ntt_maxl", maxl)
        p.set_tuning("ntt_logt12", lt)
        med, mn = timed()
        print("%-95s %.3f ms  (min %.3f)  %.1f G elems/s" % (name + ":", med, mn, W * (1 << logn) / med / 1e6), flush=True)
