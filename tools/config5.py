"""BASELINE configs[4] shape on ONE GPU, in one process: a 64-block batch of chunk STARKs through the engine, then the
multi-scalar multiplications a Groth16 wrap of a 2^26-constraint circuit performs (G1: A, B1, L, H over 2^26 points each;
G2: B over 2^26 points), on synthetic points/scalars.  The recursive-verifier circuit itself does not exist offline
(DESIGN.md par.7), so the MSMs run on random 253-bit scalars; FFTs over F_r are not included.
usage: python tools/config5.py [chunks=64] [logn=20] [msm_log=26]"""
import ctypes as C, json, os, random, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import DeviceBuffer
from eigen_zeth_amd.service import bn254
from eigen_zeth_amd.service.engine import Engine, EngineConfig
from eigen_zeth_amd.service.server import default_backend_factory

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 20
mlog = int(sys.argv[3]) if len(sys.argv) > 3 else 26
cfg = EngineConfig(air="chunk64", logn=logn, crs_dir=os.path.join(tempfile.gettempdir(), "zp_crs_c5"), witness_threads=16)
eng = Engine(default_backend_factory(0), cfg)
eng.groth16_keys()
p = eng.be.p
rnd = random.Random(3)
n = 1 << mlog
g = np.random.default_rng(7)
t1 = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in pt for k in range(8)] for pt in [bn254.g1_mul(rnd.randrange(1, bn254.R)) for _ in range(32)]], dtype=np.uint32)
t2 = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]) for k in range(8)]
               for pt in [bn254.g2_mul(rnd.randrange(1, bn254.R)) for _ in range(16)]], dtype=np.uint32)
scs = g.integers(0, 1 << 32, size=(n, 8), dtype=np.uint64).astype(np.uint32); scs[:, 7] &= 0x1FFFFFFF
d_s = DeviceBuffer(p, scs.size // 2); p._chk(p.lib.zp_h2d(p.ctx, d_s.ptr, scs.ctypes.data, scs.nbytes))
pts1 = t1[g.integers(0, 32, size=n)]
d_p1 = DeviceBuffer(p, pts1.size // 2); p._chk(p.lib.zp_h2d(p.ctx, d_p1.ptr, pts1.ctypes.data, pts1.nbytes)); del pts1
pts2 = t2[g.integers(0, 16, size=n)]
d_p2 = DeviceBuffer(p, pts2.size // 2); p._chk(p.lib.zp_h2d(p.ctx, d_p2.ptr, pts2.ctypes.data, pts2.nbytes)); del pts2
o1, o2 = (C.c_uint32 * 16)(), (C.c_uint32 * 32)()
for rep in range(2):
    t0 = time.perf_counter()
    ch = eng.gen_batch_chunks("b", list(range(1, K + 1)), 12345, "evm")
    proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    ta = time.perf_counter()
    agg = eng.aggregate("b", proofs[0]["proof"], proofs[-1]["proof"])
    fin = eng.final("b", agg, "BN128", "479881985774944702531460751064278034642760119942")
    tb = time.perf_counter()
    for _ in range(4):
        p._chk(p.lib.zp_msm_bn254(p.ctx, d_p1.ptr, d_s.ptr, n, o1))
    tc = time.perf_counter()
    p._chk(p.lib.zp_msm_bn254_g2(p.ctx, d_p2.ptr, d_s.ptr, n, o2))
    td = time.perf_counter()
    print(json.dumps({"rep": rep, "chunks": K, "logn": logn, "stark_batch_s": round(ta - t0, 3), "aggregate_final_stark_and_groth16_s": round(tb - ta, 3),
                      "msm_g1_4x_2^%d_s" % mlog: round(tc - tb, 3), "msm_g2_2^%d_s" % mlog: round(td - tc, 3), "total_s": round(td - t0, 3)}), flush=True)
