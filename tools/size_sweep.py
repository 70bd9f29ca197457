"""NTT / LDE / Merkle throughput over the size sweep of SURVEY.md 8d (N = 2^20 .. 2^26, W = 32 columns, blow-up 2),
device-resident data, HIP-side wall around sync'd calls.  usage: python tools/size_sweep.py [logn ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

W = 32
p = Prover(0)
rows = []
# integer roofline: lane-instructions per second the chip can issue (every instruction of these kernels is of the 4-cycle class) against
# the VALU counts of the committed PMC runs: 318.7 per element over the three passes of the 2^24 plan = 106.2 per pass (profiles/r5_integer_roofline.json;
# applied per pass of whatever plan a size gets: an APPROXIMATION outside 2^22 .. 2^26), 17.5 k per Poseidon permutation (profiles/r5_pmc_sq_merkle*.json)
info = p.device_info()
CEIL = info["cus"] * 4 * 64 * info["clock_khz"] * 1e3 / 4.0
VALU_PER_ELEM_PASS, VALU_PER_PERM = 318.7 / 3, 13.5e3      # (round 6, second half: 13.5 k per permutation with the blocked partial rounds, profiles/r6c_integer_roofline.json; 17.5 k before)
for logn in [int(a) for a in sys.argv[1:]] or [20, 21, 22, 23, 24, 25, 26]:
    N, M = 1 << logn, 2 << logn
    x = np.random.default_rng(logn).integers(0, 2**62, size=(W, N), dtype=np.uint64)
    d = p.upload(x); del x
    o = p.alloc(W * N); e = p.alloc(W * M); t = p.alloc((2 * M - 1) * 4)

    def timed(fn, reps=3):
        fn(); p.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        p.sync()
        return (time.perf_counter() - t0) / reps

    tn = timed(lambda: p.ntt(d, o, logn, W))
    tl = timed(lambda: p.lde(d, e, logn, 1, W))
    tm = timed(lambda: p.merkle_commit(e, M, W, t))
    perms = ((W + 7) // 8) * M + M - 1
    np_n, np_m = len(p.ntt_plan(logn)["passes"]), len(p.ntt_plan(logn + 1)["passes"])
    r = {"logn": logn, "cols": W, "passes": np_n, "ntt_ms": tn * 1e3, "ntt_Gelems_s": W * N / tn / 1e9,
         "ntt_frac_hbm": 16.0 * N * W / tn / 8e12, "ntt_frac_int": VALU_PER_ELEM_PASS * np_n * W * N / CEIL / tn,
         "lde_ms": tl * 1e3, "lde_GBs": 24.0 * N * W / tl / 1e9, "lde_frac_hbm": 24.0 * N * W / tl / 8e12,
         "lde_frac_int": VALU_PER_ELEM_PASS * (np_n * N + np_m * M) * W / CEIL / tl,
         "merkle_ms": tm * 1e3, "merkle_Gperms_s": perms / tm / 1e9, "merkle_frac_hbm": (8.0 * M * W + 32.0 * (2 * M - 1)) / tm / 8e12,
         "merkle_frac_int": VALU_PER_PERM * perms / CEIL / tm}
    rows.append(r)
    print(json.dumps(r), flush=True)
    for b in (d, o, e, t):
        b.free()
