"""world_size-2 gloo test (CPU) of the multi-GPU commit: column shards + one all-to-all + local
subtrees + all-gather must give exactly the single-process Merkle root."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, W, logm, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigen_zeth_amd import multigpu
    from eigen_zeth_amd.poseidon_constants import default_round_constants, default_mds
    from oracle import oracle as O
    rc = np.array(default_round_constants(), dtype=np.uint64)
    mds = np.array(default_mds(), dtype=np.uint64)
    M = 1 << logm
    full = O.random_field((W, M), 4242)                      # every rank derives the same matrix ...
    Wl = W // world
    local = torch.from_numpy(full[rank * Wl:(rank + 1) * Wl].view(np.int64).copy())   # ... and keeps its columns

    def commit_rows(mat):
        a = np.ascontiguousarray(mat.numpy().view(np.uint64))
        assert a.shape == (W, M // world)
        assert (a == full[:, rank * (M // world):(rank + 1) * (M // world)]).all()   # all columns, my rows
        return [int(v) for v in O.merkle_commit(a, rc, mds)[-1]]

    def hash_pair(l, r):
        st = np.array([list(l) + list(r) + [0, 0, 0, 0]], dtype=np.uint64)
        return [int(v) for v in O.poseidon_perm(st, rc, mds)[0][:4]]

    root, stats = multigpu.distributed_commit(local, commit_rows, hash_pair)
    if rank == 0:
        ref = [int(v) for v in O.merkle_commit(full, rc, mds)[-1]]
        out_q.put((root, ref, stats["sent_bytes"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,W,logm", [(2, 6, 8), (2, 16, 10)])
def test_distributed_commit_equals_single_root(world, W, logm):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, W, logm, q)) for r in range(world)]
    for p in procs:
        p.start()
    root, ref, sent = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert root == ref
    assert sent == (W // world) * (1 << logm) * 8 * (world - 1) // world


def test_pack_layout():
    from eigen_zeth_amd import multigpu
    x = torch.arange(2 * 8, dtype=torch.int64).view(2, 8)
    p = multigpu.pack_for_exchange(x, 4)
    assert p.shape == (4, 2, 2) and p[1].tolist() == [[2, 3], [10, 11]]


def _fs_worker(rank, world, port, logn, inverse, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigen_zeth_amd import multigpu
    from oracle import oracle as O
    N = 1 << logn
    full = O.random_field((1, N), 777 + logn)
    local = torch.from_numpy(full[0, rank * (N // world):(rank + 1) * (N // world)].view(np.int64).copy())

    def ntt_rows(mat, inv):
        a = np.ascontiguousarray(mat.numpy().view(np.uint64))
        return torch.from_numpy((O.intt(a) if inv else O.ntt(a)).view(np.int64))

    def twiddle_rows(mat, row0, logn_total, inv):
        a = np.ascontiguousarray(mat.numpy().view(np.uint64))
        w = O.lib().orc_root(O.ROOT32_DEFAULT, logn_total)
        if inv:
            w = pow(w, O.P - 2, O.P)
        out = np.empty_like(a)
        for r in range(a.shape[0]):
            for k in range(a.shape[1]):
                out[r, k] = int(a[r, k]) * pow(w, (row0 + r) * k, O.P) % O.P
        return torch.from_numpy(out.view(np.int64))

    got = multigpu.four_step_ntt(local, logn, ntt_rows, twiddle_rows, inverse=inverse)
    ref = (O.intt(full) if inverse else O.ntt(full))[0, rank * (N // world):(rank + 1) * (N // world)]
    ok = bool((got.numpy().view(np.uint64) == ref).all())
    t = torch.tensor([1 if ok else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out_q.put(int(t.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,logn,inverse", [(2, 8, False), (2, 9, False), (2, 8, True)])
def test_four_step_ntt_of_one_split_column_equals_the_plain_transform(world, logn, inverse):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, world, port, logn, inverse, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == 1


def _msm_worker(rank, world, port, n, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import random
    from eigen_zeth_amd import multigpu
    from oracle import naive_bn254 as B
    rnd = random.Random(5)
    pts = [B.mul(B.G, rnd.randrange(1, B.R)) for _ in range(n)]
    scs = [rnd.randrange(B.R) for _ in range(n)]
    if n >= 4:   # rank 1's range sums to infinity: the flag word must carry that
        half = n // 2
        pts[half:] = [pts[half], (pts[half][0], (-pts[half][1]) % B.Q)] + [None] * (n - half - 2)
        scs[half:] = [7, 7] + [0] * (n - half - 2)
    lo, hi = rank * n // world, (rank + 1) * n // world
    got = multigpu.distributed_msm(lambda: B.msm([p for p in pts[lo:hi]], scs[lo:hi]), B.add)
    ref = B.msm(pts, scs)
    t = torch.tensor([1 if got == ref else 0])
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        out_q.put(int(t.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_distributed_msm_equals_single_msm():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_msm_worker, args=(r, 2, port, 12, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=10) == 1
