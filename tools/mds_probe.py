import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd import native
from eigen_zeth_amd.native import Prover
from eigen_zeth_amd.poseidon_constants import default_mds
p = Prover(0)
logm, W = 22, 32
M = 1 << logm
x = np.random.default_rng(1).integers(0, 2**63, size=(W, M), dtype=np.uint64)
d = p.upload(x); t = p.alloc((2 * M - 1) * 4)
def run(label):
    p.merkle_commit(d, M, W, t); p.sync()
    t0 = time.perf_counter()
    for _ in range(3):
        p.merkle_commit(d, M, W, t)
    p.sync()
    dt = (time.perf_counter() - t0) / 3
    perms = ((W + 7) // 8) * M + (M - 1)
    print("%s: merkle 2^%d x %d: %.2f ms  %.3f G perms/s" % (label, logm, W, dt * 1e3, perms / dt / 1e9))
run("default MDS")
mds = list(default_mds()); mds[5] += 1
p.set_constants(native.ZP_CONST_POSEIDON_MDS, mds)
run("injected MDS")
