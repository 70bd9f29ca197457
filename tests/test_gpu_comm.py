"""RCCL behind the C-ABI (csrc/comm.hip: zp_comm_*, zp_exchange_columns_to_rows, zp_merkle_commit_sharded) -- SURVEY.md 8e,
BASELINE.json configs[3].  On the one-GPU box the communicator has one rank (RCCL init, grouped send/recv to self, all-gather
and broadcast all execute); with >= 2 GPUs the compiled host host/commit_sharded runs one process per GPU and every rank must
print the single-GPU root.  The multi-rank layout logic is the one the world-2 gloo tests pin (tests/test_multigpu_cpu.py)."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from eigen_zeth_amd import native
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_comm_world_of_one_collectives_and_sharded_commit(prover, tables):
    rc, mds = tables
    comm = native.Comm(prover, 0, 1, native.comm_unique_id())
    try:
        x = O.random_field((4, 1 << 10), 5)
        d_x, d_y = prover.upload(x), prover.alloc(x.size)
        comm.all_to_all(d_x, d_y, x.size)                       # one peer: the whole buffer goes to itself
        assert (prover.download(d_y, x.shape) == x).all()
        prover.memset(d_y, 0, x.size * 8)
        comm.all_gather(d_x, d_y, x.size)
        assert (prover.download(d_y, x.shape) == x).all()
        comm.broadcast(d_y, x.size, 0)
        assert (prover.download(d_y, x.shape) == x).all()
        d_pack, d_rows = prover.alloc(x.size), prover.alloc(x.size)
        comm.exchange_columns_to_rows(d_x, 4, 1 << 10, d_pack, d_rows)
        assert (prover.download(d_rows, x.shape) == x).all()
        # the sharded commitment of a world of one IS the plain commitment: whole tree and root against the oracle
        M, W = 1 << 12, 9
        cols = O.random_field((W, M), 6)
        ref = O.merkle_commit(cols, rc, mds)
        d_cols, d_tree = prover.upload(cols), prover.alloc((2 * M - 1) * 4)
        root = comm.merkle_commit_sharded(d_cols, M, W, d_tree)
        assert root == [int(v) for v in ref[-1]]
        assert (prover.download(d_tree, ref.shape) == ref).all()
        # what the TRANSPORT says about this communicator (ncclCommCount / ncclCommUserRank / ncclCommCuDevice), and the exchange buffers the
        # commitment keeps between calls (round 6: one call in a dozen took seconds while they were allocated and freed per call) can be given back
        assert comm.info() == {"transport": "rccl", "ranks_seen": 1, "user_rank": 0, "device": 0}
        comm.release_scratch()
        assert comm.merkle_commit_sharded(d_cols, M, W, d_tree) == root
    finally:
        comm.close()


def test_compiled_host_shards_one_commitment_over_the_visible_gpus(prover, tables):
    """host/commit_sharded (C++ on include/zeth_prover.h alone): one process per GPU, RCCL id through a file; every rank prints
    the root a single GPU commits.  World = the number of visible GPUs rounded down to a power of two (1 on the test box)."""
    rc, mds = tables
    exe = os.path.join(ROOT, "host", "commit_sharded")
    assert os.path.exists(exe), "build first: make -C host"
    n = native.device_count()       # not torch: its wheel carries its own librccl, a second copy in this process (double free at exit)
    world = 1
    while world * 2 <= min(n, 4):
        world *= 2
    logn, logb, W = 12, 1, 8
    x = O.random_field((W, 1 << logn), 77)
    want = O.merkle_commit(O.lde(x, logb), rc, mds)[-1]
    with tempfile.TemporaryDirectory() as td:
        x.tofile(os.path.join(td, "trace.bin"))
        idf = os.path.join(td, "rccl.id")
        with open(idf, "wb") as f:      # a stale id file of an earlier run (another nonce) must not be taken for this run's (host/rendezvous.hpp)
            f.write(b"ZPRCCLID" + (1).to_bytes(8, "little") + bytes(128))
        nonce = str(0x5EED0000 + os.getpid())
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([exe, os.path.join(td, "trace.bin"), str(logn), str(logb), str(W), str(r), str(world), idf, nonce], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
        outs = [p.communicate(timeout=300) for p in procs]
        assert not os.path.exists(idf), "rank 0 retires its id record once the communicator stands"
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
        words = so.split("root ")[1].split()[:4]
        assert [int(w, 16) for w in words] == [int(v) for v in want], so


def test_an_aborted_rccl_communicator_refuses_collectives(prover):
    """zp_comm_abort on an RCCL communicator (one rank on this box): ncclCommAbort takes it down, every later collective answers ZP_ERR_COMM
    instead of waiting for peers that will never come; zp_comm_set_timeout_ms arms the watchdog of the RCCL path (csrc/comm.hip: rccl_wait)"""
    comm = native.Comm(prover, 0, 1, native.comm_unique_id())
    try:
        comm.set_timeout_ms(5000)
        d_x, d_y = prover.upload(O.random_field((4,), 3)), prover.alloc(4)
        comm.all_gather(d_x, d_y, 4)
        assert (prover.download(d_y, (4,)) == prover.download(d_x, (4,))).all()
        comm.abort()
        with pytest.raises(native.ZpError) as e:
            comm.all_gather(d_x, d_y, 4)
        assert e.value.code == -6                 # ZP_ERR_COMM
        comm.abort()                              # idempotent
    finally:
        comm.close()
