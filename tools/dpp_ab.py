"""same-box A/B of the in-wave (DPP / ds_swizzle / ds_bpermute) forms against the product's LDS / register forms (csrc/wave_xchg.hpp):
  1. transforms of <= 4096 points (ntt_small_kernel): the last six radix-2 stages as lane exchanges (knob ntt_small_wave = 1) against LDS + one
     workgroup barrier per stage;
  2. the FRI fold by 16 (fri_fold_kernel<4>): a coset spread over the 16 lanes of a DPP row (knob fri_fold_lanes = 1) against sixteen inputs
     per lane in registers.
Outputs are compared bit for bit.  Measurement tool.  usage: python tools/dpp_ab.py > profiles/r5_dpp_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

P = (1 << 64) - (1 << 32) + 1
p = Prover(0)


def ab(knob, run, reps=21):
    res, outs = {0: [], 1: []}, {}
    for rep in range(reps):
        for k in (0, 1):
            p.set_tuning(knob, 1 if k else 2)          # 1: the in-wave form always, 2: never (0 = the default: where it measured faster)
            p.sync()
            t0 = time.perf_counter()
            outs[k] = run()
            p.sync()
            if rep:
                res[k].append((time.perf_counter() - t0) * 1e3)
    p.set_tuning(knob, 0)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}, outs


print("# tools/dpp_ab.py on one MI355X: ms per call (median of 20, alternating); 'same' = outputs equal bit for bit")
print("# 1. small transforms: W columns of 2^logn points, one workgroup per column")
rng = np.random.default_rng(1)
for logn, W in ((4, 4096), (6, 4096), (6, 65536), (8, 4096), (8, 65536), (10, 1024), (10, 16384), (12, 256), (12, 4096), (12, 16384)):
    x = rng.integers(0, P, size=(W, 1 << logn), dtype=np.uint64)
    d, o = p.upload(x), p.alloc(W << logn)
    for inverse in (False, True):
        def run():
            (p.intt if inverse else p.ntt)(d, o, logn, W)
            return None
        med, _ = ab("ntt_small_wave", run)
        got = {}
        for k in (0, 1):
            p.set_tuning("ntt_small_wave", 1 if k else 2)
            (p.intt if inverse else p.ntt)(d, o, logn, W)
            got[k] = p.download(o, (W, 1 << logn))
        p.set_tuning("ntt_small_wave", 0)
        print("%s 2^%-2d x %-6d  LDS stages %.4f ms   in-wave last six %.4f ms   (%.2fx)  same: %s"
              % ("intt" if inverse else "ntt ", logn, W, med[0], med[1], med[0] / med[1], bool((got[0] == got[1]).all())), flush=True)
    d.free(); o.free()
print("# 2. FRI fold by 16: three planes of 2^logn values -> 2^(logn-4)")
for logn in (12, 16, 20, 23):
    x = rng.integers(0, P, size=(3, 1 << logn), dtype=np.uint64)
    d, o = p.upload(x), p.alloc(3 << (logn - 4))
    beta = [int(v) for v in rng.integers(0, P, size=3, dtype=np.uint64)]

    def run():
        p.fri_fold(d, o, logn, 4, beta, 7)
        return None
    med, _ = ab("fri_fold_lanes", run)
    got = {}
    for k in (0, 1):
        p.set_tuning("fri_fold_lanes", 1 if k else 2)
        p.fri_fold(d, o, logn, 4, beta, 7)
        got[k] = p.download(o, (3, 1 << (logn - 4)))
    p.set_tuning("fri_fold_lanes", 0)
    print("fold 2^%-2d  registers (16 inputs per lane) %.4f ms   DPP rows (1 input per lane) %.4f ms   (%.2fx)  same: %s"
          % (logn, med[0], med[1], med[0] / med[1], bool((got[0] == got[1]).all())), flush=True)
    d.free(); o.free()
