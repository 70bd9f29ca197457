"""Multi-rank self-check of the sharded paths on real GPUs over RCCL (BASELINE configs[3] form).
Launch:  python -m torch.distributed.run --nnodes=1 --nproc-per-node G --master-addr 127.0.0.1 --master-port P tools/multigpu_check.py
Every rank owns one GPU.  Checked against ONE GPU (rank 0 recomputes on the gathered inputs):
  1. column shards -> LDE -> all-to-all -> row-sharded Poseidon Merkle -> all-gathered sub-roots == single-GPU root
  2. four-step NTT of one column split over the ranks == zp_ntt of the whole column
  3. MSM by point ranges + all-gather of the partial sums == single MSM
  4. one chunk STARK sharded over the ranks (stark/sharded.py) == the single-GPU proof, byte for byte
Rank 0 prints one JSON line; exit code 1 on any mismatch."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from bench import random_field_tensor
    from eigen_zeth_amd import multigpu
    from eigen_zeth_amd.native import Prover
    from eigen_zeth_amd.service import bn254

    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    # ZP_CHECK_BACKEND=gloo: rehearsal of several ranks on fewer GPUs than ranks (RCCL refuses two ranks on one device):
    # every rank computes on GPU local % device_count, the collectives are staged through the host (multigpu.py)
    backend = os.environ.get("ZP_CHECK_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    prover = Prover(local, stream=torch.cuda.current_stream().cuda_stream)
    multigpu.use_device_layout(prover)
    logn, cols = int(os.environ.get("ZP_CHECK_LOGN", "16")), 4
    N, M = 1 << logn, 2 << logn
    res, ok = {"world": world, "logn": logn, "backend": backend}, True
    u64 = lambda t: [int(v) & 0xFFFFFFFFFFFFFFFF for v in t.tolist()]

    def hash_pair_on(p, st):
        def f(l, r):
            vals = [v - (1 << 64) if v >= (1 << 63) else v for v in (list(l) + list(r) + [0] * 4)]
            st.copy_(torch.tensor(vals, dtype=torch.int64))
            p.poseidon_perm(st, 1)
            return u64(st[:4])
        return f

    # ---- 1. sharded commit
    x = random_field_tensor(torch, (cols, N), dev, 1000 + rank)           # this rank's columns
    y = torch.empty((cols, M), dtype=torch.int64, device=dev)
    prover.lde(x, y, logn, 1, cols)
    Wtot, Mloc = cols * world, M // world
    tree = torch.empty(((2 * Mloc - 1) * 4,), dtype=torch.int64, device=dev)
    st = torch.zeros((12,), dtype=torch.int64, device=dev)

    def commit_rows(mat):
        prover.merkle_commit(mat, Mloc, Wtot, tree)
        return u64(tree[-4:])
    t0 = time.perf_counter()
    root, stats = multigpu.distributed_commit(y, commit_rows, hash_pair_on(prover, st))
    torch.cuda.synchronize()
    res["commit_ms"] = (time.perf_counter() - t0) * 1e3
    allx = [torch.empty_like(x) for _ in range(world)]
    multigpu.all_gather(allx, x)
    if rank == 0:
        full = torch.cat(allx, dim=0)                                     # [Wtot][N] in rank order = column order
        yf = torch.empty((Wtot, M), dtype=torch.int64, device=dev)
        prover.lde(full, yf, logn, 1, Wtot)
        tf = torch.empty(((2 * M - 1) * 4,), dtype=torch.int64, device=dev)
        prover.merkle_commit(yf, M, Wtot, tf)
        want = u64(tf[-4:])
        res["commit_root_matches_single_gpu"] = root == want
        ok &= root == want
        del full, yf, tf
    # ---- 2. one column split over the ranks
    flog = logn + 4
    blk = random_field_tensor(torch, ((1 << flog) // world,), dev, 2000 + rank)
    out = multigpu.four_step_ntt(blk, flog, *multigpu.hip_row_ops(prover))
    parts_in = [torch.empty_like(blk) for _ in range(world)]
    parts_out = [torch.empty_like(out) for _ in range(world)]
    multigpu.all_gather(parts_in, blk)
    multigpu.all_gather(parts_out, out)
    if rank == 0:
        col = torch.cat(parts_in).view(1, -1)
        ref = torch.empty_like(col)
        prover.ntt(col, ref, flog, 1)
        same = bool(torch.equal(ref.view(-1), torch.cat(parts_out)))
        res["four_step_matches_plain_ntt"] = same
        ok &= same
    # ---- 3. MSM over point ranges
    import random as _random
    rnd = _random.Random(11)
    table = [bn254.g1_mul(rnd.randrange(1, bn254.R)) for _ in range(16)]
    tab = np.array([[(c >> (32 * k)) & 0xFFFFFFFF for c in pt for k in range(8)] for pt in table], dtype=np.uint32)
    nloc = 1 << 12
    g = [np.random.default_rng(300 + r) for r in range(world)]
    idx = [gr.integers(0, 16, size=nloc) for gr in g]
    scs = [gr.integers(0, 1 << 32, size=(nloc, 8), dtype=np.uint64).astype(np.uint32) for gr in g]
    for s in scs:
        s[:, 7] &= 0x1FFFFFFF
    add = lambda p, q: bn254._pt_add(bn254._Ops1, p, q)
    total = multigpu.distributed_msm(lambda: prover.msm_bn254_arrays(tab[idx[rank]], scs[rank]), add)
    if rank == 0:
        single = prover.msm_bn254_arrays(np.concatenate([tab[i] for i in idx]), np.concatenate(scs))
        res["msm_matches_single_gpu"] = total == single
        ok &= total == single
    # ---- 4. one proof spread over the ranks (row-sharded quotient / DEEP, gather before FRI) == the single-GPU proof
    from eigen_zeth_amd import native as _native
    from eigen_zeth_amd.stark import air as AIR, prover as PR
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    from eigen_zeth_amd.stark.sharded import HipShardOps, ShardedBackend
    air = AIR.get_air("chunk64")
    slog = int(os.environ.get("ZP_CHECK_STARK_LOGN", "14"))
    tr, pub = _native.synth_trace(air.trace_kind, slog, air.width, 31)
    params = PR.StarkParams(slog, 1, 3, 4, 8, pow_bits=8)
    t0 = time.perf_counter()
    sharded = PR.proof_to_json(PR.prove(air, tr, pub, params, ShardedBackend(HipShardOps(prover, dev))))
    res["sharded_proof_ms"] = (time.perf_counter() - t0) * 1e3
    if rank == 0:
        single = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover)))
        res["sharded_proof_matches_single_gpu"] = sharded == single
        ok &= sharded == single
    flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
    multigpu.broadcast(flag, 0)
    if rank == 0:
        res["ok"] = bool(ok)
        print(json.dumps(res), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
