"""the shape bench.py --gpus 8 runs (one 2^28-element column over 8 ranks, zp_ntt_sharded) rehearsed with 8 thread-ranks on ONE GPU
(in-process communicator) against zp_ntt of the whole column on one ctx (check tool)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
logn, G = 28, 8
N = 1 << logn
rng = np.random.default_rng(5)
x = rng.integers(0, 2**62, size=N, dtype=np.uint64)
p0 = native.Prover(0)
d = p0.upload(x); o = p0.alloc(N)
p0.ntt(d, o, logn, 1); p0.sync()
want = p0.download(o, (N,))
d.free(); o.free()
group = native.CommGroup(G)
out = [None] * G; walls = [0.0] * G
def body(r):
    p = native.Prover(0); c = native.Comm(p, r, G, group=group)
    blk = x[r * (N // G):(r + 1) * (N // G)]
    dd, t = p.upload(blk), p.alloc(2 * (N // G))
    t0 = time.perf_counter()
    c.ntt_sharded(dd, t, logn); p.sync()
    walls[r] = time.perf_counter() - t0
    out[r] = p.download(dd, (N // G,))
    c.close(); p.close()
ts = [threading.Thread(target=body, args=(r,)) for r in range(G)]
[t.start() for t in ts]; [t.join() for t in ts]
group.close()
got = np.concatenate(out)
print("2^28 column over 8 thread-ranks on one GPU: equals the single-ctx zp_ntt:", bool((got == want).all()), "max rank wall %.1f ms" % (max(walls) * 1e3))
