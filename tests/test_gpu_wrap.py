"""The Groth16 wrap on the MI355X (GenFinalProof, proto/prover/v1/prover.proto:130-148): the key's group elements from
zp_fixed_base_mul_bn254(_g2), H from zp_qap_quotient_bn254 with its coefficients left in HBM as MSM scalars, five MSMs over key points
resident in HBM -- the proof must be, byte for byte, the one the checker's trapdoor prover computes with three scalar multiplications
(oracle/groth16_trapdoor.py: same key scalars, same blinding), pass the pairing check, and its public input must be the commitment
oracle/wrap_verify.py recomputes from the final STARK.  Then the same through the engine at the service's parameters (a 2^21-domain circuit)."""
import json

import numpy as np
import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
from eigen_zeth_amd.service import groth16 as G16
from eigen_zeth_amd.service import wrap_circuit as WC
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from eigen_zeth_amd.stark.backend_hip import HipBackend
from oracle import groth16_verify as GV
from oracle import stark_verify as V
from oracle import wrap_verify as WV
from cpu_wrap_backend import CpuWrapBackend as CpuBackend

pytestmark = pytest.mark.gpu


def test_gpu_groth16_of_the_wrap_circuit_equals_the_trapdoor_proof(tables):
    bn = bn254_poseidon_params(17)
    hip = HipBackend(0, hash_mode="bn128")
    cpu = CpuBackend(*tables, hash_mode="bn128", bn_tables=bn)
    air = AIR.get_air("wide8")
    tr, pub = native.synth_trace(air.trace_kind, 8, air.width, 5)
    params = PR.StarkParams(8, 2, 3, 3, 6, pow_bits=0, hash="bn128")
    proof = json.loads(hip.prove_native(air, tr, pub, params))
    assert V.verify(proof, air.program(), *tables, V.expectation(params.to_dict()), bn)
    wc = WC.wrap_circuit(WC.Layout.of_air(air, params))
    aux = 12345
    # the caller-set wires from the prover's own binary openings record = what the Python reference assignment reads out of the proof text
    tlog = WC.TranscriptLog(proof, wc.layout, WC.head_values(air, params, proof["root32"], proof["shift"]))      # the sponge, replayed on the host
    assert tlog.indices == [q["index"] for q in proof["queries"]]
    w0, mask = wc.assign(proof, aux, tlog)
    set_idx, set_val = native.wrap_assign(wc.script, hip.stark_openings(), aux)
    # the one-call prover's own log of its transcript (absorbed blocks, final rates) = the replay, word for word
    assert (hip.stark_openings() == WC.openings_record(proof, wc.layout, tlog)).all()
    assert sorted(set_idx.tolist()) == np.flatnonzero(mask).tolist() and (w0[set_idx.astype(np.int64)] == set_val).all()
    # witness completion and A w, B w, C w on the GPU (zp_r1cs_eval_device: the gadget instances by the permutation kernel, the glue by a sparse-row
    # kernel) = the host's generic evaluation of the same blob (zp_r1cs_eval), word for word
    wf, a, b, c = native.r1cs_eval(wc.blob, w0, mask)
    gw, ga, gb, gc, gpub = hip.p.r1cs_eval_device(wc.blob, set_idx, set_val)
    assert (gw == wf).all() and (ga == a).all() and (gb == b).all() and (gc == c).all() and gpub == native.fr_ints(wf[1:2])
    with pytest.raises(native.ZpError):                                    # a wire nobody set
        hip.p.r1cs_eval_device(wc.blob, set_idx[:-1], set_val[:-1])
    other = wc.blob.copy()
    at = 16 + int(other[6]) + (int(other[6]) + 1) + int(other[16 + int(other[6]) + int(other[6])])      # first coefficient of the template's A matrix
    other[at] ^= np.uint64(2)
    with pytest.raises(native.ZpError, match="gadget"):                   # another gadget than the kernel's: refused, not mis-evaluated
        hip.p.r1cs_eval_device(other, set_idx, set_val)
    key = G16.Key(wc.blob)
    rand = (0x1234567890ABCDEF1234567890ABCDEF, 0xFEDCBA0987654321FEDCBA0987654321)
    p_gpu, pubs, ms = G16.prove(key, set_idx, set_val, hip, rand)          # ONE library call: zp_groth16_prove
    p_cpu, pubs_c, _ = G16.prove(key, set_idx, set_val, cpu, rand)
    assert pubs == pubs_c == [WV.public_input(proof, aux, bn, air.program())]
    assert p_gpu == p_cpu                                   # QAP transforms + five MSMs on the GPU = three scalar multiplications by the trapdoor
    assert WV.verify(key.vk, p_gpu, pubs, proof, aux, bn, air.program())
    assert WV.verify_rest(proof, air.program(), *tables, V.expectation(params.to_dict()), bn)
    # stage B-1: the openings of query 1 offered for query 0 (valid leaves and paths, wrong place) -- no witness
    moved = json.loads(json.dumps(proof))
    moved["queries"][0] = json.loads(json.dumps(proof["queries"][1]))
    mi, mv = native.wrap_assign(wc.script, WC.openings_record(moved, wc.layout, tlog), aux)
    with pytest.raises(ValueError, match="does not satisfy"):
        G16.prove(key, mi, mv, hip, rand)
    assert not GV.verify(key.vk, p_gpu, [(pubs[0] + 1) % G16.R])
    bad = set_val.copy()
    bad[int(np.flatnonzero(set_idx == np.uint64(wc.q[1]["trees"][2]["levels"][0]["sib"][5]))[0]), 0] ^= np.uint64(1)      # one digest of one path
    with pytest.raises(ValueError, match="does not satisfy"):
        G16.prove(key, set_idx, bad, hip, rand)
    print("zp_groth16_prove ms (witness, QAP, MSMs):", [round(x, 2) for x in ms])
    print("wrap circuit: %d constraints, domain 2^%d, %d wires" % (wc.c.n_constraints, wc.c.logm(), wc.c.n_wires))

    # ---- stage B-2 (round 6): the circuit built FOR this statement also runs the verifier's field arithmetic (two arithmetic templates: witness
    # programs on the host, their rows by a kernel); its public input commits to public data only
    from eigen_zeth_amd.service import wrap_arith as WA
    head = WC.head_values(air, params, proof["root32"], proof["shift"])
    wc2 = WC.wrap_circuit(WC.Layout.of_air(air, params), WA.Statement(air.program(), proof["root32"], proof["shift"], head))
    aux_l, zw = native.wrap_aux(hip.stark_openings(), air.program(), proof["publics"], params.logn, proof["root32"], aux)
    assert aux_l == [aux] and zw == [(tlog.chal[1] >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(3)]
    w0, mask = wc2.assign(proof, aux_l, tlog)
    set_idx, set_val = native.wrap_assign(wc2.script, hip.stark_openings(), aux_l)
    assert sorted(set_idx.tolist()) == np.flatnonzero(mask).tolist() and (w0[set_idx.astype(np.int64)] == set_val).all()
    wf, a, b, c = native.r1cs_eval(wc2.blob, w0, mask)
    gw, ga, gb, gc, gpub = hip.p.r1cs_eval_device(wc2.blob, set_idx, set_val)
    assert (gw == wf).all() and (ga == a).all() and (gb == b).all() and (gc == c).all() and gpub == native.fr_ints(wf[1:2])
    # the witness the GPU completed, judged by the checker's own reader of the blob (oracle/r1cs_blob.py: no code shared with builder, host or kernels)
    from oracle import r1cs_blob as RB
    assert RB.first_violated(wc2.blob, native.fr_ints(gw)) == -1
    key2 = G16.Key(wc2.blob)
    p_gpu, pubs, ms2 = G16.prove(key2, set_idx, set_val, hip, rand)
    p_cpu, pubs_c, _ = G16.prove(key2, set_idx, set_val, cpu, rand)
    stmt = (air.program(), params.to_dict(), proof["root32"], proof["shift"], proof["publics"])
    assert p_gpu == p_cpu and pubs == pubs_c == [WV.public_input_b2(*stmt, aux, zw, bn)]
    assert WV.verify_b2(key2.vk, p_gpu, pubs, *stmt, aux, zw, bn)
    # a STARK the GPU prover makes HONESTLY FROM A FALSE WITNESS: hashes, indices, transcript all consistent (the hashing-only circuit proves it),
    # the arithmetic is not -- no witness on the host, no proof on the GPU
    tr2 = tr.copy()
    tr2[5, 100] ^= np.uint64(1)
    bad = json.loads(hip.prove_native(air, tr2, pub, params))
    rec_bad = hip.stark_openings().copy()
    tl_bad = WC.TranscriptLog(bad, wc.layout, head)
    bi, bv = native.wrap_assign(wc.script, rec_bad, aux)
    G16.prove(key, bi, bv, hip, rand)                                     # rounds 4-5: a pairing-valid proof "of" a false statement's hashing
    aux_b, _ = native.wrap_aux(rec_bad, air.program(), bad["publics"], params.logn, bad["root32"], aux)
    bi, bv = native.wrap_assign(wc2.script, rec_bad, aux_b)
    with pytest.raises(ValueError, match="does not satisfy"):
        G16.prove(key2, bi, bv, hip, rand)
    with pytest.raises(ValueError, match="does not satisfy"):
        native.r1cs_eval(wc2.blob, *wc2.assign(bad, aux_b, tl_bad))
    # one arithmetic wire of a complete witness changed: refused by the rows kernel as by the host
    print("stage B-2: %d constraints (global template %d rows, query template %d x %d), zp_groth16_prove ms:" % (wc2.c.n_constraints, wc2.arith_stats["global_rows"],
          wc2.arith_stats["query_rows"], params.n_queries), [round(x, 2) for x in ms2])


def test_engine_final_proof_wraps_the_final_stark(tables, tmp_path):
    """GenFinalProof at the service's default parameters: the Groth16 proof verifies under the engine's key, its public input commits to THIS
    final STARK and the request's aggregator address, the same request with deterministic blinding gives the same proof.json, and a final
    STARK with a flipped digest has no witness"""
    from eigen_zeth_amd.service.engine import Engine, EngineConfig
    from eigen_zeth_amd.service.server import default_backend_factory
    from eigen_zeth_amd.service import consumer as CS
    bn = bn254_poseidon_params(17)
    cfg = EngineConfig(air="chunk64", logn=14, chunks_per_block=1, groth16_seed="test")
    eng = Engine(default_backend_factory(0), cfg)
    ch = eng.gen_batch_chunks("w", [3, 4], 12345, "evm")
    proofs = eng.gen_chunk_proofs("w", ch["task_id"], ch["chunk_count"], ch["batch_data"])
    agg = eng.aggregate("w", proofs[0]["proof"], proofs[1]["proof"])
    addr = "479881985774944702531460751064278034642760119942"
    js, pub_js = eng.final("w", agg, "BN128", addr)
    fs = json.loads(eng.final_starks["w"])
    pr = CS.parse_proof(js)
    pub = CS.parse_public_input(pub_js)
    proof = {"pi_a": tuple(pr.a), "pi_b": (pr.b.x, pr.b.y), "pi_c": tuple(pr.c)}
    vk = json.loads(eng.verifying_key_json(2, 14))
    g1 = lambda d: (int(d["x"]), int(d["y"]))
    g2 = lambda d: ((int(d["x"][0]), int(d["x"][1])), (int(d["y"][0]), int(d["y"][1])))
    vkp = {"alpha1": g1(vk["alpha1"]), "beta2": g2(vk["beta2"]), "gamma2": g2(vk["gamma2"]), "delta2": g2(vk["delta2"]), "ic": [g1(p) for p in vk["ic"]]}
    # stage B-2: everything a reader of the final proof checks -- d from PUBLIC data (the statement, its public inputs, the address, the proof text's zeta),
    # the pairing -- with no part of the final STARK but its public inputs (which are the aggregated proof's STARK's roots, indices and transcripts)
    fp = eng.final_stark_params(json.loads(agg)["stark"])
    meta = json.loads(js)
    zw = [int(v) for v in meta["zeta"]]
    stmt = (eng.final_programs["w"], fp.to_dict(), fs["root32"], fs["shift"], fs["publics"])
    assert WV.verify_b2(vkp, proof, pub, *stmt, int(addr), zw, bn)
    with pytest.raises(V.Reject):
        WV.verify_b2(vkp, proof, pub, *stmt, int(addr) + 1, zw, bn)
    assert V.verify(fs, eng.final_programs["w"], *tables, V.expectation(fp.to_dict(), fs["root32"], fs["shift"]), bn)     # (the STARK the circuit verified does verify)
    assert "stage B-2" in meta["circuit"] and "2^22" in meta["circuit"] and "test key" in meta["circuit"]
    print("wrap info:", json.dumps(eng.wrap_info))
    js2, pub2 = eng.final("w", agg, "BN128", addr)
    assert js2 == js and pub2 == pub_js                       # deterministic blinding: the same proof.json
    js3, pub3 = eng.final("w", agg, "BN128", "1")
    assert pub3 != pub_js                                     # another aggregator address: another statement
    assert any(k.startswith("final/verify-aggregated-header") for k in eng.stage_timings["final/w"])
    # an aggregated proof whose STARK fails what NO QUERY covers is refused before it is wrapped: a final layer that is not low degree passes every
    # fold of the verifier AIR (the last fold is compared with the layer as given), so only the native header check (stark/verifier.py) stands
    # in its way -- round 4's engine wrapped such a text into a pairing-valid proof
    from eigen_zeth_amd.stark import verifier as SV
    bad = json.loads(agg)
    with pytest.raises(ValueError):            # (a text altered by hand also changes its transcript, so the witness builder may speak first; a
        b2 = json.loads(agg)                   # consistently forged proof -- false trace, honest transcript -- is what only the header check stops)
        b2["stark"]["evals"]["z"][0][0] ^= 1
        eng.final("w", json.dumps(b2, separators=(",", ":")), "BN128", addr)
    # ... and the same holds ONE LEVEL DOWN (round-5 advisor item): the chunk-proof headers inside the aggregated proof are verified too -- identity at
    # zeta, final layer, grinding -- and the aggregation STARK's public inputs must be what THOSE headers dictate.  A header whose out-of-domain
    # evaluation was changed, or swapped for the header of another honest proof, is refused although the aggregation STARK itself is untouched.
    for mutate in (lambda a: a["inner"][0]["evals"]["z"][5].__setitem__(0, a["inner"][0]["evals"]["z"][5][0] ^ 1),
                   lambda a: a["inner"][1]["fri"]["final"][0].__setitem__(2, a["inner"][1]["fri"]["final"][0][2] ^ 1),
                   lambda a: a["inner"].__setitem__(0, a["inner"][1])):
        b3 = json.loads(agg)
        mutate(b3)
        with pytest.raises(ValueError):
            eng.final("w", json.dumps(b3, separators=(",", ":")), "BN128", addr)
    ap = eng._agg_params(eng._own_shape(bad))
    from eigen_zeth_amd.stark import verifier_air as VA
    vair = VA.verifier_air(eng._own_shape(bad), *eng._tables(eng.be))
    assert SV.verify_header(bad["stark"], vair, ap, eng.be)["indices"] == [q["index"] for q in bad["stark"]["queries"]]
    bad["stark"]["evals"]["zw"][3][1] ^= 1
    with pytest.raises(SV.Reject, match="identity"):
        SV.verify_header(bad["stark"], vair, ap, eng.be)
    print("final stage timings:", json.dumps({k: round(v * 1e3, 1) for k, v in eng.stage_timings["final/w"].items()}))
