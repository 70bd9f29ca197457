"""same-box A/B of the extension (blow-up 2) at the provers' sizes: the fused seam kernel on the seam plans (knob lde_seam_plans = 1, round 5)
against the unfused two-transform path on the default plans (0), without the coefficient store (the provers' call since round 5) and with
it (rounds 1-4's call).  Measurement tool.  usage: python tools/lde_seam_plans_ab.py [cols] > profiles/r5_lde_seam_plans_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
rng = np.random.default_rng(5)
print("# tools/lde_seam_plans_ab.py on one MI355X: ms per extension, median of 7 (alternating), W columns")
for logn in (16, 18, 19, 20, 21, 22, 23, 24):
    W = int(sys.argv[1]) if len(sys.argv) > 1 else max(8, min(76, (1 << 28) >> logn))
    x = rng.integers(0, 2**63, size=(W, 1 << logn), dtype=np.uint64)
    d_in, d_out, d_coef = p.upload(x), p.alloc(W << (logn + 1)), p.alloc(W << logn)
    res = {}
    for rep in range(8):
        for name, knob, coef in (("seam-plans", 1, None), ("default-plans", 0, None), ("default-plans+coef", 0, d_coef)):
            p.set_tuning("lde_seam_plans", knob)
            p.sync()
            t0 = time.perf_counter()
            p.lde(d_in, d_out, logn, 1, W, d_coef=coef)
            p.sync()
            if rep:
                res.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    p.set_tuning("lde_seam_plans", 1)
    plan = p.ntt_plan(logn)["lde"]
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print("2^%d x %d  %s  | seam-plans %.3f ms  default-plans %.3f ms  default-plans+coefficient-store %.3f ms  | fused %s inv %s fwd %s | %.0f GB/s algorithmic"
          % (logn, W, "", med["seam-plans"], med["default-plans"], med["default-plans+coef"], plan["seam_fused"], plan["inverse_radix_logs"],
             plan["forward_radix_logs"], 24.0 * W * (1 << logn) / med["seam-plans"] / 1e6), flush=True)
    for d in (d_in, d_out, d_coef):
        d.free()
