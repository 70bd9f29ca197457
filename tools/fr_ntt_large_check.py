"""F_r NTT at 2^24 and 2^26 points: direct evaluation of outputs of a sparse input + round trip (measurement / check tool)"""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eigen_zeth_amd.native import Prover
from oracle import naive as NV
p = Prover(0)
for logn in (24, 26):
    n = 1 << logn
    rng = np.random.default_rng(5)
    sparse = np.zeros((n, 4), dtype=np.uint64)
    pos = sorted(set(int(v) for v in rng.integers(0, n, size=16)))
    vals = {}
    for i in pos:
        w = rng.integers(0, 1 << 62, size=4, dtype=np.uint64); w[3] &= np.uint64((1 << 60) - 1)
        sparse[i] = w; vals[i] = sum(int(w[k]) << (64 * k) for k in range(4))
    d = p.upload(sparse.reshape(-1))
    t0 = time.perf_counter(); p.ntt_bn254(d, logn, False, 7); p.sync(); dt = time.perf_counter() - t0
    w = NV.fr_root(logn)
    ok = True
    for k in (0, 1, 12345, n // 2 + 3, n - 1):
        z = 7 * pow(w, k, NV.FR) % NV.FR
        want = sum(vals[i] * pow(z, i, NV.FR) for i in pos) % NV.FR
        got = p._fr_ints(p.download(d, (1, 4), offset_elems=k * 4))[0]
        ok &= got == want
    p.ntt_bn254(d, logn, True, 7)
    back = p.download(d, (n, 4))
    ok &= bool((back == sparse).all())
    print("F_r NTT 2^%d: %.1f ms (incl. first-use tables), evaluations + round trip ok = %s" % (logn, dt * 1e3, ok), flush=True)
    d.free()
