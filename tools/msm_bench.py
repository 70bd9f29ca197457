"""BN254 MSM throughput on one GPU (measurement tool): points drawn from a 64-point table"""
import os, sys, time, random, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover, DeviceBuffer
from eigen_zeth_amd.service import bn254
p = Prover(0)
if os.environ.get('ZP_MSM_C'): p.set_tuning('msm_c', int(os.environ['ZP_MSM_C']))
rnd = random.Random(1)
table = [bn254.g1_mul(rnd.randrange(1, bn254.R)) for _ in range(64)]
tab = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in pt for k in range(8)] for pt in table], dtype=np.uint32)
for logn in [int(a) for a in sys.argv[1:]] or [16, 20, 22]:
    n = 1 << logn
    g = np.random.default_rng(logn)
    pts = tab[g.integers(0, 64, size=n)]
    scs = g.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32); scs[:, 7] &= 0x1FFFFFFF   # uniform 253-bit scalars (< r): the top window is a few heavy buckets
    d_p, d_s = DeviceBuffer(p, pts.size // 2), DeviceBuffer(p, scs.size // 2)
    p._chk(p.lib.zp_h2d(p.ctx, d_p.ptr, pts.ctypes.data, pts.nbytes)); p._chk(p.lib.zp_h2d(p.ctx, d_s.ptr, scs.ctypes.data, scs.nbytes))
    out = (C.c_uint32 * 16)()
    p._chk(p.lib.zp_msm_bn254(p.ctx, d_p.ptr, d_s.ptr, n, out))
    t0 = time.perf_counter()
    for _ in range(2):
        p._chk(p.lib.zp_msm_bn254(p.ctx, d_p.ptr, d_s.ptr, n, out))
    dt = (time.perf_counter() - t0) / 2
    print("MSM n=2^%d: %.2f ms  %.2f M points/s  alg %.1f GB/s" % (logn, dt * 1e3, n / dt / 1e6, 96.0 * n / dt / 1e9), flush=True)
    d_p.free(); d_s.free()
# G2 (the B element of Groth16): ZP_MSM_G2="18 20" python tools/msm_bench.py 0
for logn in [int(a) for a in os.environ.get("ZP_MSM_G2", "").split()]:
    n = 1 << logn
    tab2 = [bn254.g2_mul(rnd.randrange(1, bn254.R)) for _ in range(16)]
    t2 = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]) for k in range(8)] for pt in tab2], dtype=np.uint32)
    g = np.random.default_rng(logn)
    pts = t2[g.integers(0, 16, size=n)]
    scs = g.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32); scs[:, 7] &= 0x1FFFFFFF
    d_p, d_s = DeviceBuffer(p, pts.size // 2), DeviceBuffer(p, scs.size // 2)
    p._chk(p.lib.zp_h2d(p.ctx, d_p.ptr, pts.ctypes.data, pts.nbytes)); p._chk(p.lib.zp_h2d(p.ctx, d_s.ptr, scs.ctypes.data, scs.nbytes))
    out = (C.c_uint32 * 32)()
    p._chk(p.lib.zp_msm_bn254_g2(p.ctx, d_p.ptr, d_s.ptr, n, out))
    t0 = time.perf_counter()
    p._chk(p.lib.zp_msm_bn254_g2(p.ctx, d_p.ptr, d_s.ptr, n, out))
    dt = time.perf_counter() - t0
    print("G2 MSM n=2^%d: %.2f ms  %.2f M points/s" % (logn, dt * 1e3, n / dt / 1e6), flush=True)
    d_p.free(); d_s.free()
