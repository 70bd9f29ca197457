"""zp_stark_prove -- the whole chunk STARK behind one C-ABI call -- against the Python orchestration over the same library:
the proof text must be byte-identical, and it must pass the independent verifier."""
import json

import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR

pytestmark = pytest.mark.gpu


def _both(prover, name, logn, params):
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    air = AIR.get_air(name)
    tr, pub = AIR.cubic_witness(logn, 11) if name == "cubic" else native.synth_trace(air.trace_kind, logn, air.width, 11)
    ref = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover, quotient="program")))
    d_tr = prover.upload(tr)
    got = prover.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pub], params.logn, params.logb, params.fri_logf,
                             params.fri_final_log, params.n_queries, params.pow_bits)
    d_tr.free()
    return air, ref, got


@pytest.mark.parametrize("name,logn,logb,logf,final,nq,pow_bits", [
    ("fib", 6, 1, 3, 3, 5, 0), ("wide8", 10, 2, 3, 4, 9, 6), ("perm", 8, 1, 2, 3, 6, 4), ("chunk16", 9, 1, 3, 3, 17, 8),
    ("cubic", 8, 2, 3, 3, 6, 0), ("chunk64", 12, 1, 3, 5, 80, 12), ("wide32", 11, 1, 4, 2, 8, 0)])
def test_native_prover_writes_the_same_proof(prover, tables, name, logn, logb, logf, final, nq, pow_bits):
    from oracle import stark_verify as V
    params = PR.StarkParams(logn, logb, logf, final, nq, pow_bits=pow_bits)
    air, ref, got = _both(prover, name, logn, params)
    assert got == ref
    assert V.verify(json.loads(got), air.program(), *tables, V.expectation(params.to_dict()))


def test_native_prover_2_20_verifies(prover, tables):
    """the service's shape: chunk AIR, 2^20 rows, 100-bit parameters, one call"""
    from oracle import stark_verify as V
    air = AIR.get_air("chunk64")
    tr, pub = native.synth_trace(air.trace_kind, 20, air.width, 5)
    d_tr = prover.upload(tr)
    params = PR.StarkParams(20, 1, 3, 5, 80, pow_bits=20)
    js = prover.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pub], 20, 1, 3, 5, 80, 20)
    d_tr.free()
    assert V.verify(json.loads(js), air.program(), *tables, V.expectation(params.to_dict()))


def test_native_prover_rejects_bad_arguments(prover):
    import numpy as np
    air = AIR.get_air("fib")
    tr, pub = native.synth_trace(air.trace_kind, 6, air.width, 1)
    d_tr = prover.upload(tr)
    prog = np.array(air.program(), dtype=np.uint64)
    with pytest.raises(native.ZpError):
        prover.stark_prove("fib", prog[:-1], d_tr, [int(v) for v in pub], 6, 1, 3, 3, 5, 0)       # truncated program
    with pytest.raises(native.ZpError):
        prover.stark_prove("fib", prog, d_tr, [int(v) for v in pub][:-1], 6, 1, 3, 3, 5, 0)       # wrong number of publics
    with pytest.raises(native.ZpError):
        prover.stark_prove("fib", prog, d_tr, [int(v) for v in pub], 6, 0, 3, 3, 5, 0)            # no blow-up
    with pytest.raises(native.ZpError):
        prover.stark_prove('fi"b', prog, d_tr, [int(v) for v in pub], 6, 1, 3, 3, 5, 0)           # a name that would break the JSON
    bad = prog.copy(); bad[0] ^= 1
    with pytest.raises(native.ZpError):
        prover.stark_prove("fib", bad, d_tr, [int(v) for v in pub], 6, 1, 3, 3, 5, 0)             # bad magic
    cub = AIR.get_air("cubic")                     # degree-3 constraints: two quotient pieces need blow-up >= 2 ... and get it
    trc, pubc = AIR.cubic_witness(6, 1)
    d_c = prover.upload(trc)
    assert AIR.quotient_chunks(cub) == 2
    prover.stark_prove("cubic", cub.program(), d_c, [int(v) for v in pubc], 6, 1, 3, 3, 5, 0)
    d_tr.free(); d_c.free()


def test_native_prover_follows_the_configured_constants(tables):
    """another 2^32-th root of unity, another coset shift and an injected (non-default) MDS: the one-call prover and the Python
    orchestration still write the same proof, and the verifier -- told the same constants -- accepts it"""
    import numpy as np
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    from oracle import stark_verify as V
    rc, mds = tables
    p = native.Prover(0)
    try:
        mds2 = np.array(mds, dtype=np.uint64).copy()
        mds2[5] = (int(mds2[5]) + 3)              # no longer the compiled-in matrix: the LDS-staged generic path
        p.set_constants(native.ZP_CONST_ROOT32, [native.ROOT32_ALT])
        p.set_constants(native.ZP_CONST_COSET_SHIFT, [7])
        p.set_constants(native.ZP_CONST_POSEIDON_MDS, mds2)
        air = AIR.get_air("chunk16")
        tr, pub = native.synth_trace(air.trace_kind, 8, air.width, 3)
        params = PR.StarkParams(8, 1, 3, 3, 7, pow_bits=5)
        ref = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=p, quotient="program")))
        d_tr = p.upload(tr)
        got = p.stark_prove(air.name, air.program(), d_tr, [int(v) for v in pub], 8, 1, 3, 3, 7, 5)
        d_tr.free()
        assert got == ref
        proof = json.loads(got)
        assert proof["root32"] == native.ROOT32_ALT and proof["shift"] == 7
        assert V.verify(proof, air.program(), rc, mds2, V.expectation(params.to_dict(), root32=native.ROOT32_ALT, shift=7))
        with pytest.raises(V.Reject):
            V.verify(proof, air.program(), rc, mds2, V.expectation(params.to_dict()))      # a verifier on the default domain refuses it
    finally:
        p.close()


def test_compiled_host_proves_a_chunk_through_the_c_abi(tmp_path, prover, tables):
    """host/prove_chunk.cpp -- a compiled host that uses nothing but include/zeth_prover.h (no Python, no torch): files in, proof
    out; the proof is the text the Python orchestration writes and the independent verifier accepts it"""
    import os
    import subprocess
    import numpy as np
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    from oracle import stark_verify as V
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "prove_chunk")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(root, "host"), "-s"])
    air = AIR.get_air("chunk16")
    tr, pub = native.synth_trace(air.trace_kind, 10, air.width, 21)
    params = PR.StarkParams(10, 1, 3, 3, 12, pow_bits=6)
    np.asarray(air.program(), dtype=np.uint64).tofile(tmp_path / "program.bin")
    np.ascontiguousarray(tr).tofile(tmp_path / "trace.bin")
    np.asarray(pub, dtype=np.uint64).tofile(tmp_path / "publics.bin")
    out = tmp_path / "proof.json"
    r = subprocess.run([exe, str(tmp_path / "program.bin"), str(tmp_path / "trace.bin"), str(tmp_path / "publics.bin"), "10", "1", "3", "3", "12", "6",
                        str(out), air.name], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    text = out.read_text()
    assert text == PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover, quotient="program")))
    assert V.verify(json.loads(text), air.program(), *tables, V.expectation(params.to_dict()))


def test_compiled_host_shards_one_proof_over_the_visible_gpus(tmp_path, prover, tables):
    """host/prove_chunk with rank / world / id-file: one process per GPU on an RCCL communicator, zp_stark_prove_sharded; rank 0's
    proof file must be the single-GPU proof text.  World = the visible GPUs rounded down to a power of two that divides the
    column count (1 on the one-GPU box: RCCL with one rank; the multi-rank logic runs as threads in
    tests/test_gpu_sharded_native.py)."""
    import os
    import subprocess
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "host", "prove_chunk")
    air = AIR.get_air("chunk16")
    tr, pub = native.synth_trace(air.trace_kind, 10, air.width, 23)
    world = 1
    while world * 2 <= min(native.device_count(), 4):
        world *= 2
    np.asarray(air.program(), dtype=np.uint64).tofile(tmp_path / "program.bin")
    np.ascontiguousarray(tr).tofile(tmp_path / "trace.bin")
    np.asarray(pub, dtype=np.uint64).tofile(tmp_path / "publics.bin")
    out, idf = tmp_path / "proof.json", tmp_path / "rccl.id"
    idf.write_bytes(bytes(128))          # a stale id file in the old bare format: ignored (host/rendezvous.hpp)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, str(tmp_path / "program.bin"), str(tmp_path / "trace.bin"), str(tmp_path / "publics.bin"), "10", "1", "3", "3", "12",
                               "6", str(out), air.name, str(r), str(world), str(idf), str(0xC0DE0000 + os.getpid())], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so + se
    d = prover.upload(tr)
    single = prover.stark_prove(air.name, air.program(), d, [int(v) for v in pub], 10, 1, 3, 3, 12, 6)
    d.free()
    assert out.read_text() == single


def test_compiled_host_answers_gen_aggregated_proof_like_the_service(tmp_path, tables):
    """host/aggregate (C++ on include/zeth_prover.h alone: zp_proof_queries_scan / _parse, zp_program_digest, zp_recursion_witness,
    zp_stark_prove; the verifier AIR and its witness schedule as data files from tools/export_recursion_shape.py) writes, for the two
    chunk proofs of a GenAggregatedProof request (prover.proto:115-126), BYTE FOR BYTE the aggregated proof the Python service's engine
    answers -- and the checker accepts it"""
    import os
    import subprocess
    import sys
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from eigen_zeth_amd.stark import verifier_air as VA
    from oracle import aggregate_verify as AV
    from oracle import stark_verify as V
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import export_recursion_shape as EX
    rc, mds = tables
    cfg = EngineConfig(air="chunk64", logn=12, chunks_per_block=1, crs_dir=str(tmp_path / "crs"), n_queries=24, pow_bits=8, agg_queries=10)
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("b", [7, 8], 12345, "evm")
    proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    want = eng.aggregate("batch-7", proofs[0]["proof"], proofs[1]["proof"])
    shape, vair, ap = EX.export(str(tmp_path / "shape"), logn=12, n_proofs=2, cfg=cfg)
    (tmp_path / "p1.json").write_text(proofs[0]["proof"])
    (tmp_path / "p2.json").write_text(proofs[1]["proof"])
    exe = os.path.join(root, "host", "aggregate")
    r = subprocess.run([exe, str(tmp_path / "shape"), "batch-7", str(tmp_path / "p1.json"), str(tmp_path / "p2.json"), str(tmp_path / "agg.json")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = (tmp_path / "agg.json").read_text()
    assert got == want
    agg = json.loads(got)
    assert AV.verify(agg, AIR.get_air("chunk64").program(), vair.program(), rc, mds, V.expectation(eng.stark_params(12).to_dict()),
                     V.expectation(ap.to_dict()), shape.n_slots())
    # a proof of the request tampered with: the compiled host refuses as the service does (no accepting witness)
    bad = json.loads(proofs[1]["proof"])
    bad["queries"][3]["quotient"]["values"][1] ^= 1
    (tmp_path / "p2.json").write_text(json.dumps(bad, separators=(",", ":")))
    r = subprocess.run([exe, str(tmp_path / "shape"), "batch-7", str(tmp_path / "p1.json"), str(tmp_path / "p2.json"), str(tmp_path / "agg2.json")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no accepting witness" in r.stderr and not (tmp_path / "agg2.json").exists()


def test_compiled_host_answers_gen_final_proof_like_the_service(tmp_path, tables):
    """host/aggregate final (C++ on include/zeth_prover.h alone: zp_recursion_witness one level up, zp_stark_prove_bn128, zp_stark_openings,
    zp_wrap_assign, zp_groth16_prove over key files) writes, for a GenFinalProof request (prover.proto:130-148) under the service's
    deterministic blinding, BYTE FOR BYTE the proof and public input the Python service's engine answers -- and they pass the pairing check and
    the recomputation of the public input from the final STARK"""
    import os
    import subprocess
    import sys
    from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
    from eigen_zeth_amd.service import consumer as CS
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from oracle import wrap_verify as WV
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import export_recursion_shape as EX
    cfg = EngineConfig(air="chunk64", logn=12, chunks_per_block=1, n_queries=24, pow_bits=8, agg_queries=10, final_queries=6, groth16_seed="host-test")
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("b", [7, 8], 12345, "evm")
    proofs = eng.gen_chunk_proofs("b", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    agg = eng.aggregate("b", proofs[0]["proof"], proofs[1]["proof"])
    addr = "479881985774944702531460751064278034642760119942"
    want_js, want_pub = eng.final("b", agg, "BN128", addr)
    wc, key, fp = EX.export_final(str(tmp_path / "final"), eng, n_proofs=2, logn=12)
    (tmp_path / "agg.json").write_text(agg)
    exe = os.path.join(root, "host", "aggregate")
    args = [exe, "final", str(tmp_path / "final"), str(tmp_path / "agg.json"), addr, "host-test", str(tmp_path / "proof.json"), str(tmp_path / "public.json")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    print(r.stdout.strip())
    assert (tmp_path / "proof.json").read_text() == want_js and (tmp_path / "public.json").read_text() == want_pub
    pr = CS.parse_proof(want_js)
    proof = {"pi_a": tuple(pr.a), "pi_b": (pr.b.x, pr.b.y), "pi_c": tuple(pr.c)}
    fs = json.loads(eng.final_starks["b"])
    assert WV.verify_b2(key.vk, proof, CS.parse_public_input(want_pub), eng.final_programs["b"], fp.to_dict(), fs["root32"], fs["shift"], fs["publics"], int(addr),
                        [int(v) for v in json.loads(want_js)["zeta"]], bn254_poseidon_params(17))
    # an aggregated proof tampered with has no witness one level up: the host refuses as the service does
    bad = json.loads(agg)
    bad["stark"]["queries"][2]["trace"]["values"][5] ^= 1
    (tmp_path / "agg.json").write_text(json.dumps(bad, separators=(",", ":")))
    r = subprocess.run(args[:6] + [str(tmp_path / "proof2.json"), str(tmp_path / "public2.json")], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and not (tmp_path / "proof2.json").exists()


def test_one_call_prover_through_the_generated_constraint_kernel(prover, tables):
    """zp_stark_set_air_kernel: with the AIR's generated kernel registered the one-call prover writes the proof text it writes through the
    interpreter; NULL forgets the kernel; a program it was not registered for is not affected"""
    from eigen_zeth_amd.stark import air as AIR
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    be = HipBackend(prover=prover)
    for name, logn in (("chunk64", 12), ("wide8", 10), ("perm", 9), ("fib", 9)):
        air = AIR.get_air(name)
        tr, pub = native.synth_trace(air.trace_kind, logn, air.width, 5)
        d = prover.upload(tr)
        args = (air.name, air.program(), d, [int(v) for v in pub], logn, 1, 3, 3, 10, 4)
        prover.set_air_kernel(air.program(), None)
        a = prover.stark_prove(*args)
        prover.set_air_kernel(air.program(), be._airlib(air))
        b = prover.stark_prove(*args)
        prover.set_air_kernel(air.program(), None)
        c = prover.stark_prove(*args)
        d.free()
        assert a == b == c
