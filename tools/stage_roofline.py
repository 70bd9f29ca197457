"""Per-stage measurement for every hot-path row of SURVEY.md 8a (N1..N6 + DEEP/OOD): HIP-side time
(wall around sync'd calls, 3 reps after a warm-up), algorithmic bytes per SURVEY 8d and fraction of the
8 TB/s HBM roof.  Run directly it measures the GPU only; the CPU restatement column is added when bench.py calls
run() from its cpu_baseline leg (`python bench.py --stage-roofline`), which is the only place that may hand the
oracle modules in.  usage: python tools/stage_roofline.py [logn=22] [W=32]"""
import json, os, sys, time, random
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # idle OpenMP workers of the CPU checker must not spin beside the GPU timing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
from eigen_zeth_amd.native import Prover
from eigen_zeth_amd.poseidon_constants import default_round_constants, default_mds
from eigen_zeth_amd.stark import air as AIR, field as F
from eigen_zeth_amd.stark.backend_hip import HipBackend

PEAK = 8000.0
P = 0xFFFFFFFF00000001


def _field(rng, shape):
    """uniform-ish canonical field elements (62-bit, < p)"""
    return rng.integers(0, 2**62, size=shape, dtype=np.uint64)


def run(logn=22, W=32, cpu=None):
    """cpu: None, or a namespace with the oracle modules (O, CpuBackend, B1) supplied by bench.py's cpu_baseline leg"""
    clogn = min(logn, 18)          # CPU sample size
    p = Prover(0)
    be = HipBackend(prover=p)
    rc, mds = np.array(default_round_constants(), dtype=np.uint64), np.array(default_mds(), dtype=np.uint64)
    O = cpu.O if cpu else None
    cbe = cpu.CpuBackend(rc, mds) if cpu else None
    N, M = 1 << logn, 1 << (logn + 1)
    rng = np.random.default_rng(1)
    out = {"logn": logn, "W": W, "stages": {}}
    if cpu:
        out["cpu_sample_logn"], out["cpu_threads"] = clogn, O.num_threads()

    def timed(fn, reps=3):
        fn(); p.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        p.sync()
        return (time.perf_counter() - t0) / reps

    def cpu_timed(fn):
        if not cpu:
            return None
        t0 = time.perf_counter(); fn(); return time.perf_counter() - t0

    def rec(name, t, alg_bytes, unit_count, unit, cpu_t, cpu_units, note=""):
        d = {"gpu_ms": t * 1e3, "algorithmic_GB": alg_bytes / 1e9, "achieved_GBs": alg_bytes / t / 1e9,
             "frac_of_hbm_peak": alg_bytes / t / 1e9 / PEAK, "gpu_rate": unit_count / t, "unit": unit, "note": note}
        if cpu_t:
            d.update({"cpu_rate": cpu_units / cpu_t, "cpu_s": cpu_t, "speedup_vs_cpu_port": (unit_count / t) / (cpu_units / cpu_t)})
        out["stages"][name] = d
        print(name, json.dumps(d), file=sys.stderr, flush=True)

    x = rng.integers(0, 2**63, size=(W, N), dtype=np.uint64)
    xc = _field(np.random.default_rng(3), (W, 1 << clogn))
    d_x = p.upload(x)
    d_ext, d_coef = p.alloc(W * M), p.alloc(W * N)
    # N1
    t = timed(lambda: p.ntt(d_x, d_coef, logn, W))
    rec("N1 ntt", t, 16.0 * N * W, W * N, "field-elems/s", cpu_timed(lambda: O.ntt(xc)), W << clogn)
    # N2
    t = timed(lambda: p.lde(d_x, d_ext, logn, 1, W, d_coef=d_coef))
    rec("N2 lde(b=2)", t, 8.0 * N * 3 * W, W * N, "trace-elems/s", cpu_timed(lambda: O.lde(xc, 1)), W << clogn)
    # N3
    d_tree = p.alloc((2 * M - 1) * 4)
    t = timed(lambda: p.merkle_commit(d_ext, M, W, d_tree))
    perms = ((W + 7) // 8) * M + M - 1
    ce = O.lde(xc, 1) if cpu else None
    cM = 2 << clogn
    rec("N3 poseidon merkle", t, 8.0 * M * W + 32.0 * (2 * M - 1), perms, "permutations/s",
        cpu_timed(lambda: O.merkle_commit(ce, rc, mds)), ((W + 7) // 8) * cM + cM - 1, "integer-VALU bound (about 25.7k VALU instructions per permutation)")
    # N4 (generated kernel for the wide AIR of this width, if built-in)
    name = "wide%d" % W
    if name in AIR.BUILTIN_AIRS:
        air = AIR.get_air(name)
        tr, pub = native.synth_trace(air.trace_kind, logn, W, 5)
        c1 = be.commit_trace(tr, logn, 1)
        fixed = be.fixed_ext(logn, 1)
        apow = [[k + 1, k + 2, k + 3] for k in range(len(air.constraints))]
        wl = F.inv(F.root(logn, be.root32))
        hold = {}
        def q():
            if "q" in hold: hold["q"].free()
            hold["q"] = be.quotient(air, c1, fixed, pub, apow, [3, 5], logn, 1, wl)
        t = timed(q)
        if cpu:
            trc, pubc = native.synth_trace(air.trace_kind, clogn, W, 5)
            cc1 = cbe.commit_trace(trc, clogn, 1)
            cfx = cbe.fixed_ext(clogn, 1)
            wlc = F.inv(F.root(clogn, be.root32))
            cbe.quotient(air, cc1, cfx, pubc, apow, [3, 5], clogn, 1, wlc)
        rec("N4 constraint quotient (%s)" % name, t, 8.0 * M * (W + 2) + 24.0 * M, M, "LDE-rows/s",
            cpu_timed(lambda: cbe.quotient(air, cc1, cfx, pubc, apow, [3, 5], clogn, 1, wlc)), 2 << clogn,
            "each row also reads row+blowup (served by L2)")
        # OOD evaluation + DEEP
        z = [5, 6, 7]
        t = timed(lambda: p.ood_eval(c1.ext, M, 2, W, logn, 49, z, want_next=True))
        rec("OOD evaluation (all columns at zeta and zeta w, from the extension: zp_ood_eval)", t, 8.0 * N * W, N * W, "rows x columns/s",
            cpu_timed(lambda: O.poly_eval_e3_cols(cc1.coef, z)), W << clogn, "the CPU leg evaluates the coefficient form at one point")
        ev = _field(np.random.default_rng(8), (W + 3, 3)); ev2 = _field(np.random.default_rng(9), (W, 3))
        d_q = hold["q"]; d_f = p.alloc(3 * M)
        t = timed(lambda: p.deep_quotient(c1.ext, W, d_q, 3, logn + 1, W, z, [1, 2, 3], [4, 5, 6], ev, ev2, 49, d_f))
        qc = _field(np.random.default_rng(10), (3, 2 << clogn))
        rec("DEEP quotient", t, 8.0 * M * (W + 3) + 24.0 * M, M, "LDE-rows/s",
            cpu_timed(lambda: O.deep_quotient(cc1.ext, qc, W, z, [1, 2, 3], [4, 5, 6], ev, ev2, fast=True)), 2 << clogn)
        # N5
        d_o = p.alloc(3 * (M >> 3))
        t = timed(lambda: p.fri_fold(d_f, d_o, logn + 1, 3, [9, 8, 7], 49))
        fc = _field(np.random.default_rng(11), (3, 2 << clogn))
        rec("N5 fri fold (8-to-1)", t, 24.0 * M * (1 + 1 / 8), M, "layer-elems/s", cpu_timed(lambda: O.fri_fold(fc, 3, [9, 8, 7], 49)), 2 << clogn)
    # N6
    from eigen_zeth_amd.service import bn254
    rnd = random.Random(1)
    table = [bn254.g1_mul(rnd.randrange(1, bn254.R)) for _ in range(64)]
    tab = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in pt for k in range(8)] for pt in table], dtype=np.uint32)
    n = 1 << min(logn, 22)
    pts = tab[rng.integers(0, 64, size=n)]
    scs = rng.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32); scs[:, 7] &= 0x1FFFFFFF
    import ctypes as C
    d_p, d_s = native.DeviceBuffer(p, pts.size // 2), native.DeviceBuffer(p, scs.size // 2)
    p._chk(p.lib.zp_h2d(p.ctx, d_p.ptr, pts.ctypes.data, pts.nbytes)); p._chk(p.lib.zp_h2d(p.ctx, d_s.ptr, scs.ctypes.data, scs.nbytes))
    o16 = (C.c_uint32 * 16)()
    t = timed(lambda: p._chk(p.lib.zp_msm_bn254(p.ctx, d_p.ptr, d_s.ptr, n, o16)), reps=2)
    cn = 64
    ct = cpu_timed(lambda: cpu.B1.msm(table[:cn], [rnd.randrange(cpu.B1.R) for _ in range(cn)]))
    rec("N6 bn254 msm", t, 96.0 * n, n, "points/s", ct, cn, "CPU column = definition-level Python double-and-add (single thread), not a Pippenger port")
    return out


if __name__ == "__main__":
    res = run(int(sys.argv[1]) if len(sys.argv) > 1 else 22, int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    print(json.dumps(res))
