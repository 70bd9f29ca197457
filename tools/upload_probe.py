import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eigen_zeth_amd.native import Prover
p = Prover(0)
x = np.random.default_rng(1).integers(0, 2**63, size=(64, 1 << 20), dtype=np.uint64)
d = p.upload(x); p.sync()
for _ in range(3):
    t0 = time.perf_counter(); p.h2d(d, x); p.sync(); dt = time.perf_counter() - t0
    print("H2D 512 MiB pageable: %.2f ms  %.1f GB/s" % (dt * 1e3, x.nbytes / dt / 1e9))
import torch
xt = torch.from_numpy(x.view(np.int64)).pin_memory()
dt_ = torch.empty_like(xt, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); dt_.copy_(xt, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("H2D 512 MiB pinned (torch): %.2f ms  %.1f GB/s" % (dt * 1e3, x.nbytes / dt / 1e9))
t0 = time.perf_counter(); o = p.download(d, x.shape); dt = time.perf_counter() - t0
print("D2H 512 MiB pageable: %.2f ms" % (dt * 1e3))
