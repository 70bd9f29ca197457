"""same-box A/B of the synthetic witness's expansion in HBM (csrc/synth.hip, zp_synth_trace_device): one 8-byte store per lane and row (round 4;
knob synth_rowwise = 1) against rows staged through LDS and written as column runs (round 5: knob 2; the default, 0, takes them from 2^21 rows).  Same trace, word for word.
Measurement tool.  usage: python tools/synth_fill_ab.py > profiles/r5_synth_fill_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
print("# tools/synth_fill_ab.py on one MI355X: ms per zp_synth_trace_device call from checkpoints (median of 10, alternating)")
for kind, logn, W in ((3, 20, 64), (3, 22, 64), (1, 22, 64), (1, 20, 200)):
    ck = p.synth_checkpoints(kind, logn, W, [7])
    out = p.alloc(W << logn)
    res, got = {0: [], 1: []}, {}
    for rep in range(11):
        for knob in (1, 0):
            p.set_tuning("synth_rowwise", 1 if knob else 2)
            p.sync()
            t0 = time.perf_counter()
            p.synth_trace_device(kind, logn, W, 7, ckpt=ck, out=out)
            p.sync()
            if rep:
                res[knob].append((time.perf_counter() - t0) * 1e3)
            elif logn <= 20:
                got[knob] = p.download(out, (W, 1 << logn))
    p.set_tuning("synth_rowwise", 0)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    same = bool((got[0] == got[1]).all()) if got else "(not downloaded)"
    print("kind %d  2^%d x %-3d (%.2f GiB): row-wise stores %.3f ms   LDS-staged column runs %.3f ms   (%.2fx)  same: %s"
          % (kind, logn, W, (W << logn) * 8 / 2**30, med[1], med[0], med[1] / med[0], same), flush=True)
    out.free(); ck.free()
