"""world_size-2 gloo test (CPU) of one proof spread over ranks past the commitment stage:
row-sharded quotient with a blow-up halo, row-sharded quotient commitment and DEEP, gather before FRI.
The sharded proof must be byte-identical to the single-process proof and pass the independent verifier."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, logn, logb, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, HERE)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from eigen_zeth_amd import native
    from eigen_zeth_amd.poseidon_constants import default_round_constants, default_mds
    from eigen_zeth_amd.stark import air as AIR, prover as PR
    from eigen_zeth_amd.stark.sharded import ShardedBackend
    from shard_ops_cpu import CpuShardOps
    rc, mds = default_round_constants(), default_mds()
    air = AIR.get_air(name)
    tr, pub = AIR.cubic_witness(logn, 4040) if name == "cubic" else native.synth_trace(air.trace_kind, logn, air.width, 4040)
    params = PR.StarkParams(logn, logb, 2, 3, 5, pow_bits=4)
    proof = PR.prove(air, tr, pub, params, ShardedBackend(CpuShardOps(rc, mds)))
    if rank == 0:
        out_q.put(PR.proof_to_json(proof))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,logn,logb", [("wide8", 7, 1), ("chunk16", 7, 1), ("fib", 6, 2), ("cubic", 6, 1)])
def test_two_rank_sharded_proof_equals_the_single_rank_proof(tables, name, logn, logb):
    import json
    from eigen_zeth_amd import native
    from eigen_zeth_amd.stark import air as AIR, prover as PR
    from oracle import stark_verify as V
    from oracle.stark_cpu import CpuBackend
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, logn, logb, q)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rc, mds = tables
    air = AIR.get_air(name)
    tr, pub = AIR.cubic_witness(logn, 4040) if name == "cubic" else native.synth_trace(air.trace_kind, logn, air.width, 4040)
    params = PR.StarkParams(logn, logb, 2, 3, 5, pow_bits=4)
    single = PR.proof_to_json(PR.prove(air, tr, pub, params, CpuBackend(rc, mds)))
    assert sharded == single
    assert V.verify(json.loads(sharded), air.program(), rc, mds, V.expectation(params.to_dict()))
