"""Poseidon-BN254 (t = 17) lane-per-permutation kernel: partial rounds in blocks of four against round by round (knob p254_block), same box
usage: python tools/p254_block_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover

p = Prover(0)
p.install_poseidon_bn254(17)
rng = np.random.default_rng(1)


def timed(fn, reps=3):
    fn(); p.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    p.sync()
    return (time.perf_counter() - t0) / reps


cnt = 1 << 18
w = rng.integers(0, 1 << 60, size=(cnt * 17, 4), dtype=np.uint64).reshape(-1)
M, W = 1 << 20, 47
cols = p.upload(rng.integers(0, 1 << 63, size=(W, M), dtype=np.uint64).reshape(-1))
tree = p.alloc(p.merkle16_nodes(M) * 4)
out = {}
for rep in range(2):
    for knob, name in ((0, "blocks of four"), (2, "round by round")):
        p.set_tuning("p254_block", knob)
        st = p.upload(w)
        dt = timed(lambda: p._chk(p.lib.zp_poseidon_bn254_perm(p.ctx, st.ptr, cnt, 17)))
        p._chk(p.lib.zp_poseidon_bn254_perm(p.ctx, st.ptr, cnt, 17))
        dtm = timed(lambda: p.merkle16_commit_bn254(cols, M, W, tree), reps=2)
        root = p.download(tree, p.merkle16_nodes(M) * 4)[-4:].tolist()
        out.setdefault(name, root)
        print("%-15s: 2^18 permutations %.3f ms = %.2f M perms/s; 16-ary commit 2^20 rows x %d columns %.2f ms; root %016x.." %
              (name, dt * 1e3, cnt / dt / 1e6, W, dtm * 1e3, root[0]), flush=True)
        st.free()
assert out["blocks of four"] == out["round by round"], "roots differ"
print("roots identical")
