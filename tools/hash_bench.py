"""Poseidon / Merkle throughput on one GPU (measurement tool)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
logm, W = 22, 32
M = 1 << logm
x = np.random.default_rng(1).integers(0, 2**63, size=(W, M), dtype=np.uint64)
d = p.upload(x); t = p.alloc((2 * M - 1) * 4)
p.merkle_commit(d, M, W, t); p.sync()
t0 = time.perf_counter()
for _ in range(3):
    p.merkle_commit(d, M, W, t)
p.sync()
dt = (time.perf_counter() - t0) / 3
perms = ((W + 7) // 8) * M + (M - 1)
print("merkle 2^%d x %d: %.2f ms  %.3f G perms/s  leaf-read %.0f GB/s" % (logm, W, dt * 1e3, perms / dt / 1e9, 8.0 * M * W / dt / 1e9))
st = p.upload(np.random.default_rng(2).integers(0, 2**63, size=(1 << 22, 12), dtype=np.uint64))
p.poseidon_perm(st, 1 << 22); p.sync()
t0 = time.perf_counter()
for _ in range(3):
    p.poseidon_perm(st, 1 << 22)
p.sync()
dt = (time.perf_counter() - t0) / 3
print("perm batch 2^22: %.2f ms  %.3f G perms/s" % (dt * 1e3, (1 << 22) / dt / 1e9))
