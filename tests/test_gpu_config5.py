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


def test_64_block_batch_at_full_size_through_the_engine(tables, tmp_path):
    """BASELINE.json configs[4] on one GPU AS STATED: 64 blocks, one chunk each at BASELINE.md's C3 / C5 shape -- 2^22 rows x 64 columns (+ 12
    stage-2 columns), blow-up 2, the service's 100-bit parameters -- through the engine in ONE batch (round 4 ran this size only inside
    bench.py's probe; the tests had 64 x 2^18 and 4 x 2^22).  Every one of the 64 proofs is bound to its block; a sample of eight -- the first and
    the last (what the client forwards) and six drawn by a fixed seed -- goes through the independent verifier; then GenAggregatedProof and
    GenFinalProof over the two ends.  A second, small batch checks that a replay is byte-identical."""
    import random
    import time
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from eigen_zeth_amd.stark import air as AIR
    rc, mds = tables
    cfg = EngineConfig(air="chunk64", logn=22, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), witness_threads=8)
    assert cfg.n_queries * cfg.logb + cfg.pow_bits >= 100          # the service default security level
    eng = Engine(default_backend_factory(0), cfg)
    blocks = list(range(1, 65))
    ch = eng.gen_batch_chunks("c5", blocks, 12345, "evm")
    assert ch["chunk_count"] == 64 and len(ch["pre_state_root"]) == 32 and len(ch["post_state_root"]) == 32
    t0 = time.perf_counter()
    proofs = eng.gen_chunk_proofs("c5", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    t_batch = time.perf_counter() - t0
    assert [p["chunk_id"] for p in proofs] == list(range(64))
    prog = AIR.get_air("chunk64").program()
    exp = V.expectation(eng.stark_params(22).to_dict())
    sample = sorted({0, 63} | set(random.Random(0xC5).sample(range(1, 63), 6)))
    assert len(sample) == 8
    seen = set()
    for i in range(64):
        pr = json.loads(proofs[i]["proof"])
        assert pr["params"]["logn"] == 22 and pr["chunk"]["block"] == blocks[i] and pr["chunk"]["chunk"] == 0
        st = pr["chunk"]["statement"]     # the proof is bound to its block: its leading publics are the statement's limbs
        assert ST.bound_to(pr, 12345, blocks[i], 0, 1, bytes.fromhex(st["pre_state_root"]), bytes.fromhex(st["post_state_root"]))
        assert not ST.bound_to(pr, 12345, blocks[i] + 1, 0, 1, bytes.fromhex(st["pre_state_root"]), bytes.fromhex(st["post_state_root"]))
        seen.add(tuple(pr["roots"]["trace"]))
        if i in sample:
            assert V.verify(pr, prog, rc, mds, exp), i
    assert len(seen) == 64                                          # different blocks, different witnesses
    t0 = time.perf_counter()
    agg = eng.aggregate("c5", proofs[0]["proof"], proofs[-1]["proof"])
    final, pub = eng.final("c5", agg, "BN128", "479881985774944702531460751064278034642760119942")
    t_rec = time.perf_counter() - t0
    fp = json.loads(final)
    for k in ("pi_a", "pi_b", "pi_c"):
        assert k in fp
    assert int(json.loads(pub)[0]) < B.R
    print("configs[4] on one GPU, as a test: 64 chunk proofs of 2^22 rows in %.2f s, aggregation + final (first use: wrap key made) %.2f s" % (t_batch, t_rec))
    del proofs
    small = Engine(default_backend_factory(0), EngineConfig(air="chunk64", logn=16, chunks_per_block=1, witness_threads=4))
    ch = small.gen_batch_chunks("r", [5, 6, 7, 8], 12345, "evm")
    first = small.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    assert small.gen_chunk_proofs("r", ch["task_id"], ch["chunk_count"], ch["batch_data"]) == first   # replay is identical


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
