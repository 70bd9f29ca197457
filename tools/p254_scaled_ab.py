"""same-box A/B of the cooperative Poseidon-BN254 kernels (t = 17: 17 lanes per permutation -- transcript steps, tree levels below 2^14 nodes, small
leaf sets) on the SCALED sparse partial rounds (round 5: three dependent Montgomery products per round on lane 0's chain) against the unscaled form
(five: x^2, x^4, x^5, m00 x^5, and the row sum reduced by a product with 1; knob p254_scaled = 2).  Same digests.  Measurement tool.
usage: python tools/p254_scaled_ab.py > profiles/r5_p254_scaled_ab.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
p.install_poseidon_bn254(17)
rng = np.random.default_rng(3)


def ab(run, reps=21):
    res, out = {0: [], 2: []}, {}
    for rep in range(reps):
        for k in (2, 0):
            p.set_tuning("p254_scaled", k)
            p.sync()
            t0 = time.perf_counter()
            out[k] = run()
            p.sync()
            if rep:
                res[k].append((time.perf_counter() - t0) * 1e3)
    p.set_tuning("p254_scaled", 0)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}, out


print("# tools/p254_scaled_ab.py on one MI355X: ms per call (median of 20, alternating)")
for count in (1, 3, 48, 1024, 8192):
    st = rng.integers(0, 1 << 60, size=(count, 17, 4), dtype=np.uint64)
    st[:, :, 3] >>= 8
    d = p.upload(st)

    def run():
        p.h2d(d, st)
        p._chk(p.lib.zp_poseidon_bn254_perm(p.ctx, d.ptr, count, 17))
        return p.download(d, st.shape)
    med, out = ab(run)
    print("%5d permutations (cooperative kernel): unscaled %.3f ms   scaled %.3f ms   (%.2fx)  same: %s"
          % (count, med[2], med[0], med[2] / med[0], bool((out[0] == out[2]).all())), flush=True)
    d.free()
for logm, W in ((8, 47), (12, 47), (12, 12), (16, 3)):
    M = 1 << logm
    cols = rng.integers(0, (1 << 64) - (1 << 32), size=(W, M), dtype=np.uint64)
    d, tree = p.upload(cols), p.alloc(p.merkle16_nodes(M) * 4)

    def run():
        p.merkle16_commit_bn254(d, M, W, tree)
        return p.download(tree, (p.merkle16_nodes(M), 4))
    med, out = ab(run)
    print("16-ary tree over 2^%d leaves of %d values: unscaled %.3f ms   scaled %.3f ms   (%.2fx)  same: %s"
          % (logm, W, med[2], med[0], med[2] / med[0], bool((out[0] == out[2]).all())), flush=True)
    d.free(); tree.free()
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd import native
air = AIR.get_air("chunk64")
for logn in (14, 18):
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 5)
    d = p.upload(tr)

    def run():
        return p.stark_prove_bn128(air.name, air.program(), d, [int(v) for v in pub], logn, 2, 4, 5, 50)
    med, out = ab(run, reps=7)
    print("BN128-mode STARK, chunk64 2^%d rows, blow-up 4, 50 queries: unscaled %.2f ms   scaled %.2f ms   (%.2fx)  same text: %s"
          % (logn, med[2], med[0], med[2] / med[0], out[0] == out[2]), flush=True)
    d.free()
