"""Checks that need torch's CUDA tensors next to the C-ABI library in ONE process (torch must initialise the GPU before
libzethprover.so is loaded, so pytest runs this file as a fresh process): on one GPU
  1. four-step NTT through the layout kernels (zp_transpose) == zp_ntt, forward and inverse
  2. the multi-GPU proof orchestration (stark/sharded.py) with its HIP ops at world size 1 == the plain HIP backend
Prints one JSON line; exit code 1 on a mismatch."""
import json
import os
import socket
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    torch.cuda.init()
    import torch.distributed as dist
    from eigen_zeth_amd import multigpu, native
    from eigen_zeth_amd.native import Prover
    from eigen_zeth_amd.stark import air as AIR, prover as PR
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    from eigen_zeth_amd.stark.sharded import HipShardOps, ShardedBackend
    dev = torch.device("cuda", 0)
    p = Prover(0, stream=torch.cuda.current_stream().cuda_stream)
    res, ok = {}, True
    logn = 20
    g = torch.Generator(device="cpu").manual_seed(7200)
    x = (torch.randint(0, 1 << 62, (1 << logn,), dtype=torch.int64, generator=g)).to(dev)
    got = multigpu.four_step_ntt(x, logn, *multigpu.hip_row_ops(p))
    ref = torch.empty_like(x)
    p.ntt(x, ref, logn, 1)
    back = multigpu.four_step_ntt(got, logn, *multigpu.hip_row_ops(p), inverse=True)
    torch.cuda.synchronize()
    res["four_step_matches_plain_ntt"] = bool(torch.equal(got, ref))
    res["four_step_inverse_round_trip"] = bool(torch.equal(back, x))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    same = True
    for name, ln in (("chunk16", 9), ("wide32", 10)):
        air = AIR.get_air(name)
        tr, pub = native.synth_trace(air.trace_kind, ln, air.width, 77)
        params = PR.StarkParams(ln, 1, 3, 4, 6, pow_bits=5)
        a = PR.proof_to_json(PR.prove(air, tr, pub, params, ShardedBackend(HipShardOps(p, dev))))
        b = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=p)))
        same &= a == b
    dist.destroy_process_group()
    res["sharded_backend_world1_matches_plain_backend"] = same
    ok = all(res.values())
    res["ok"] = ok
    print(json.dumps(res), flush=True)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
