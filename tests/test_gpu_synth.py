"""zp_synth_trace_device / zp_synth_checkpoints (csrc/synth.hip): the traces of the host generator zp_synth_trace_bound, word for word."""
import numpy as np
import pytest

from eigen_zeth_amd import native

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001


@pytest.mark.parametrize("kind,logn,W,bind", [
    (0, 10, 2, None), (0, 13, 2, [5, P - 1]),
    (1, 9, 3, None), (1, 13, 17, [1, 2, 3, 4]), (1, 12, 70, [7]), (1, 6, 130, None),
    (2, 11, 3, None),
    (3, 8, 12, None), (3, 13, 24, [11, 12, 13, 14, 15, 16]), (3, 14, 76, [P - 1, 0, 1, 2, 3]), (3, 17, 76, [1, 2, 3, 4, 5, 6]), (3, 1, 16, None),
])
def test_device_trace_equals_host_trace(prover, kind, logn, W, bind):
    """both expansion kernels (knob synth_rowwise: 1 = one store per lane and row, 2 = rows staged through LDS, written as column runs; the
    default takes the second from 2^21 rows: profiles/r5_synth_fill_ab.txt)"""
    seed = 0xC0FFEE + 17 * logn + W
    tr, pub = native.synth_trace(kind, logn, W, seed, bind=bind)
    for knob in (1, 2):
        prover.set_tuning("synth_rowwise", knob)
        try:
            d, dpub = prover.synth_trace_device(kind, logn, W, seed, bind=bind)
            got = prover.download(d, (W, 1 << logn))
        finally:
            prover.set_tuning("synth_rowwise", 0)
        d.free()
        assert (dpub == pub).all()
        assert (got == tr).all()


def test_batch_checkpoints_then_traces(prover):
    kind, logn, W = 3, 14, 76
    seeds = [100, 200, 300]
    binds = [[1, 2, 3, 4, 5, 6], [7, 8, 9, 10, 11, 12], [P - 1, P - 2, 0, 0, 1, 1]]
    ck = prover.synth_checkpoints(kind, logn, W, seeds, binds)
    for i in (2, 0, 1):
        tr, pub = native.synth_trace(kind, logn, W, seeds[i], bind=binds[i])
        d, dpub = prover.synth_trace_device(kind, logn, W, seeds[i], bind=binds[i], ckpt=ck, ckpt_index=i)
        assert (dpub == pub).all() and (prover.download(d, (W, 1 << logn)) == tr).all()
        d.free()
    ck.free()


def test_bad_arguments(prover):
    with pytest.raises(native.ZpError):
        prover.synth_trace_device(3, 10, 11, 1)          # chunk AIR needs 12 columns
    with pytest.raises(native.ZpError):
        prover.synth_trace_device(0, 10, 2, 1, bind=[P])  # not canonical
    with pytest.raises(native.ZpError):
        prover.synth_trace_device(2, 10, 3, 1, bind=[1])  # kind 2 takes no bind
    with pytest.raises(ValueError):
        prover.synth_checkpoints(0, 10, 2, [1])           # no recurrence column


def test_engine_proofs_from_device_witnesses_equal_those_from_host_witnesses():
    """the same batch through the engine with witness = "host" and witness = "device" (the first witness_threads chunks come from the host
    generator either way, the others from checkpoints): the same proof texts"""
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    texts = {}
    for mode in ("host", "device"):
        eng = Engine(default_backend_factory(0), EngineConfig(air="chunk64", logn=13, chunks_per_block=3, witness_threads=2, prover_streams=4, witness=mode))
        ch = eng.gen_batch_chunks("b", [5, 6, 7], 12345, "evm")
        assert ch["chunk_count"] == 9
        proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
        texts[mode] = [p["proof"] for p in proofs]
    assert texts["host"] == texts["device"]
