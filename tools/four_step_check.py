"""four-step NTT of one column through the HIP kernels on ONE GPU (G = 1: the transposes are local), checked against
zp_ntt / zp_intt of the whole column.  Run as its own process: torch is imported before the library so that both
share one HIP runtime.  usage: python tools/four_step_check.py [logn ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eigen_zeth_amd import multigpu
from eigen_zeth_amd.native import Prover

dev = torch.device("cuda", 0)
p = Prover(0, stream=torch.cuda.current_stream().cuda_stream)
ops = multigpu.hip_row_ops(p)
g = torch.Generator(device=dev); g.manual_seed(5)
for logn in [int(a) for a in sys.argv[1:]] or [16, 21, 24]:
    N = 1 << logn
    x = torch.randint(0, 1 << 62, (1, N), dtype=torch.int64, device=dev, generator=g)   # canonical (< p)
    for inverse in (False, True):
        ref = torch.empty_like(x)
        (p.intt if inverse else p.ntt)(x, ref, logn, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        got = multigpu.four_step_ntt(x.reshape(-1).clone(), logn, *ops, inverse=inverse)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        ok = bool(torch.equal(got, ref.reshape(-1)))
        print("four-step logn=%d inverse=%d %s  %.2f ms" % (logn, inverse, "OK" if ok else "MISMATCH", dt * 1e3), flush=True)
        if not ok:
            sys.exit(1)
print("ALL OK")
