"""NTT pass plans compared on one GPU: largest radix per pass 2^9 (three passes from 2^19) vs 2^10 .. 2^12 (two passes up to
2^20 / 2^22 / 2^24).  Checks that every plan gives bit-identical output, then times them.  (measurement tool)
usage: python tools/ntt_plan_sweep.py [logn ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

logs = [int(a) for a in sys.argv[1:]] or [18, 19, 20]
maxls = [int(v) for v in os.environ.get("ZP_MAXLS", "9,10").split(",")]
p = Prover(0)
for logn in logs:
    cols = max(1, min(64, (1 << 29) >> logn))
    N = 1 << logn
    x = np.random.default_rng(logn).integers(0, 2**63, size=(cols, N), dtype=np.uint64)
    d = p.upload(x)
    o = p.alloc(cols * N)
    ref = {}
    for maxl in maxls:
        p.set_tuning("ntt_maxl", maxl)
        for inverse in (False, True):
            f = p.intt if inverse else p.ntt
            f(d, o, logn, cols); p.sync()
            got = p.download(o, (cols, N))[: 2]
            if maxl == maxls[0]:
                ref[inverse] = got
            elif not (got == ref[inverse]).all():
                print("MISMATCH: plan with max radix 2^%d differs (logn %d, inverse %s): %d of %d elements" % (
                    maxl, logn, inverse, int((got != ref[inverse]).sum()), got.size), flush=True)
            t0 = time.perf_counter()
            for _ in range(10):
                f(d, o, logn, cols)
            p.sync()
            dt = (time.perf_counter() - t0) / 10
            print("logn %2d cols %3d max radix 2^%-2d %s: %8.3f ms  %6.1f Gelem/s  plan %s" % (
                logn, cols, maxl, "intt" if inverse else "ntt ", dt * 1e3, cols * N / dt / 1e9,
                [ps["radix_log"] for ps in p.ntt_plan(logn)["passes"]]), flush=True)
    d.free(); o.free()
