"""BASELINE.json configs[4]: a 64-block batch of chunk STARKs through the engine + the wrap's 2^26-point BN254 MSM,
on one MI355X (the 8-GPU form shards the same work by chunk / by point range with no exchange: tests/test_gpu_multirank.py,
tests/test_multigpu_cpu.py)."""
import json

import numpy as np
import pytest

from oracle import naive_bn254 as B
from oracle import oracle as O
from oracle import stark_verify as V
from oracle import statement as ST

pytestmark = pytest.mark.gpu


def test_64_block_batch_through_the_engine(tables, tmp_path):
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from eigen_zeth_amd.stark import air as AIR
    rc, mds = tables
    cfg = EngineConfig(air="chunk64", logn=18, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), witness_threads=8)
    assert cfg.n_queries * cfg.logb + cfg.pow_bits >= 100          # the service default security level
    eng = Engine(default_backend_factory(0), cfg)
    blocks = list(range(1, 65))
    ch = eng.gen_batch_chunks("c5", blocks, 12345, "evm")
    assert ch["chunk_count"] == 64 and len(ch["pre_state_root"]) == 32 and len(ch["post_state_root"]) == 32
    proofs = eng.gen_chunk_proofs("c5", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    assert [p["chunk_id"] for p in proofs] == list(range(64))
    prog = AIR.get_air("chunk64").program()
    exp = V.expectation(eng.stark_params(18).to_dict())
    seen = set()
    for i in (0, 17, 63):                                           # first, last (what the client forwards) and one inside
        pr = json.loads(proofs[i]["proof"])
        assert pr["chunk"]["block"] == blocks[i] and pr["chunk"]["chunk"] == 0
        st = pr["chunk"]["statement"]     # the proof is bound to its block: its leading publics are the statement's limbs
        assert ST.bound_to(pr, 12345, blocks[i], 0, 1, bytes.fromhex(st["pre_state_root"]), bytes.fromhex(st["post_state_root"]))
        assert not ST.bound_to(pr, 12345, blocks[i] + 1, 0, 1, bytes.fromhex(st["pre_state_root"]), bytes.fromhex(st["post_state_root"]))
        assert V.verify(pr, prog, rc, mds, exp)
        seen.add(tuple(pr["roots"]["trace"]))
    assert len(seen) == 3                                           # different blocks, different witnesses
    assert eng.gen_chunk_proofs("c5", ch["task_id"], ch["chunk_count"], ch["batch_data"]) == proofs   # replay is identical
    agg = eng.aggregate("c5", proofs[0]["proof"], proofs[-1]["proof"])
    final, pub = eng.final("c5", agg, "BN128", "479881985774944702531460751064278034642760119942")
    fp = json.loads(final)
    for k in ("pi_a", "pi_b", "pi_c"):
        assert k in fp
    assert int(json.loads(pub)[0]) < B.R


@pytest.mark.parametrize("logn", [22, 26])
def test_msm_of_distinct_points_against_the_discrete_log_identity(prover, logn):
    """n DISTINCT points P_i = (1025 + i) G (oracle/bn254_gen.c) with uniform 253-bit scalars:
    sum_i s_i P_i = (sum_i s_i (1025 + i) mod r) G.  logn = 26 is the BASELINE configs[4] size (4 GiB of points)."""
    n = 1 << logn
    pts = O.bn254_consecutive_points(n, 1025)
    g = np.random.default_rng(26)
    scs = np.empty((n, 8), dtype=np.uint32)
    step = 1 << 22
    for i in range(0, n, step):
        scs[i:i + step] = g.integers(0, 1 << 32, size=(min(step, n - i), 8), dtype=np.uint32)
    scs[:, 7] &= 0x1FFFFFFF
    scs[:5] = 0
    scs[5, 0] = 1                                                   # zero scalars and a one among the first few
    want = B.mul(B.G, O.bn254_weighted_scalar_sum(scs, 1025) % B.R)
    assert prover.msm_bn254_arrays(pts, scs) == want


def test_chunks_at_the_full_c3_shape_through_the_engine(tables, tmp_path):
    """BASELINE.md's C3/C5 chunk shape: 2^22 rows x 64 columns (+ 12 stage-2 columns), blow-up 2, at the service's
    100-bit parameters -- four of configs[4]'s 64 chunks at FULL size through the engine (the 64-chunk test above runs
    2^18-row chunks); every proof is checked by the independent verifier and a replay is byte-identical"""
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from eigen_zeth_amd.stark import air as AIR
    rc, mds = tables
    cfg = EngineConfig(air="chunk64", logn=22, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), witness_threads=8,
                       prover_streams=2)
    assert cfg.n_queries * cfg.logb + cfg.pow_bits >= 100
    eng = Engine(default_backend_factory(0), cfg)
    blocks = [7, 8, 9, 10]
    ch = eng.gen_batch_chunks("c5full", blocks, 12345, "evm")
    assert ch["chunk_count"] == 4
    proofs = eng.gen_chunk_proofs("c5full", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    prog = AIR.get_air("chunk64").program()
    exp = V.expectation(eng.stark_params(22).to_dict())
    roots = set()
    for i in range(4):
        pr = json.loads(proofs[i]["proof"])
        assert pr["params"]["logn"] == 22 and pr["chunk"]["block"] == blocks[i]
        assert V.verify(pr, prog, rc, mds, exp)
        roots.add(tuple(pr["roots"]["trace"]))
    assert len(roots) == 4
    assert eng.gen_chunk_proofs("c5full", ch["task_id"], ch["chunk_count"], ch["batch_data"]) == proofs
