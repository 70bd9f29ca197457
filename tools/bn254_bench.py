"""BN128-side throughput on one GPU (measurement tool): F_r NTT sizes, the QAP quotient (seven transforms + pointwise),
Poseidon-BN254 permutations (t = 17) and the 16-ary Merkle commit.  usage: python tools/bn254_bench.py [logn ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
logs = [int(a) for a in sys.argv[1:]] or [16, 20, 22, 24]
rng = np.random.default_rng(1)


def timed(fn, reps=3):
    fn(); p.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    p.sync()
    return (time.perf_counter() - t0) / reps


for logn in logs:
    n = 1 << logn
    w = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64); w[:, 3] &= np.uint64((1 << 60) - 1)
    d = p.upload(w.reshape(-1))
    dt = timed(lambda: p.ntt_bn254(d, logn, False, None))
    dtc = timed(lambda: p.ntt_bn254(d, logn, True, 7))
    print("F_r NTT 2^%d: %.3f ms forward (%.2f G elems/s, %.0f GB/s of 64 B/elem/pass-free accounting), %.3f ms coset inverse" %
          (logn, dt * 1e3, n / dt / 1e9, 64.0 * n / dt / 1e9, dtc * 1e3), flush=True)
    if logn <= 24:
        b, c = p.upload(w.reshape(-1)), p.upload(w.reshape(-1))
        cw = p._fr_words([7])
        dq = timed(lambda: p._chk(p.lib.zp_qap_quotient_bn254(p.ctx, d.ptr, b.ptr, c.ptr, logn, cw.ctypes.data)), reps=2)
        print("QAP quotient 2^%d (3 iNTT + 3 coset NTT + pointwise + coset iNTT): %.3f ms" % (logn, dq * 1e3), flush=True)
        b.free(); c.free()
    d.free()

p.install_poseidon_bn254(17)
cnt = 1 << 16
st = p.upload(rng.integers(0, 1 << 60, size=(cnt * 17, 4), dtype=np.uint64).reshape(-1))
dt = timed(lambda: p._chk(p.lib.zp_poseidon_bn254_perm(p.ctx, st.ptr, cnt, 17)))
print("Poseidon-BN254 t=17: %d permutations in %.3f ms = %.2f M perms/s" % (cnt, dt * 1e3, cnt / dt / 1e6), flush=True)
M, W = 1 << 18, 48
cols = p.upload(rng.integers(0, 1 << 63, size=(W, M), dtype=np.uint64).reshape(-1))
tree = p.alloc(p.merkle16_nodes(M) * 4)
dt = timed(lambda: p.merkle16_commit_bn254(cols, M, W, tree), reps=2)
print("merkle16 BN254 2^18 rows x %d columns: %.2f ms (%.2f M perms/s)" % (W, dt * 1e3, (M + M // 15) / dt / 1e6), flush=True)
