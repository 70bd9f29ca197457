"""BN128-hash mode of the STARK (the last STARK before the Groth16 wrap, prover.proto:130-148): 16-ary Poseidon-BN254
Merkle trees + a transcript over the BN254 scalar field.  CPU: orchestration on the checker's backend + the independent
verifier; GPU: the proof from the MI355X is byte-identical and verifies."""
import copy
import json

import pytest

from eigen_zeth_amd import native
from eigen_zeth_amd.poseidon_constants import bn254_poseidon_params
from eigen_zeth_amd.stark import air as AIR
from eigen_zeth_amd.stark import prover as PR
from oracle import stark_verify as V
from oracle.stark_cpu import CpuBackend


@pytest.fixture(scope="module")
def bn_tables():
    return bn254_poseidon_params(17)


def _case(name, logn, seed=5):
    air = AIR.get_air(name)
    tr, pub = native.synth_trace(air.trace_kind, logn, air.width, seed)
    return air, tr, pub


@pytest.mark.parametrize("name,logn,logb", [("fib", 6, 1), ("chunk16", 7, 1), ("wide8", 6, 2)])
def test_bn128_cpu_proof_verifies_and_tampering_is_rejected(tables, bn_tables, name, logn, logb):
    be = CpuBackend(*tables, hash_mode="bn128", bn_tables=bn_tables)
    air, tr, pub = _case(name, logn)
    params = PR.StarkParams(logn, logb, 2, 3, 5, hash="bn128")
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, be)))
    expect = V.expectation(params.to_dict())
    assert expect["hash"] == "bn128" and len(proof["roots"]["trace"]) == 1
    assert V.verify(proof, air.program(), *tables, expect, bn_tables)
    # a Goldilocks-mode verifier refuses it, and a relabelled proof does not pass either
    gl = dict(expect); gl.pop("hash")
    with pytest.raises(V.Reject):
        V.verify(proof, air.program(), *tables, gl, bn_tables)
    for mutate in (lambda p: p["queries"][0]["trace"]["values"].__setitem__(0, (p["queries"][0]["trace"]["values"][0] + 1) % V.P),
                   lambda p: p["queries"][1]["trace"]["path"][0].__setitem__(3, str((int(p["queries"][1]["trace"]["path"][0][3]) + 1) % V.R_BN254)),
                   lambda p: p["roots"].__setitem__("quotient", [str((int(p["roots"]["quotient"][0]) + 1) % V.R_BN254)]),
                   lambda p: p["fri"]["final"][0].__setitem__(0, (p["fri"]["final"][0][0] + 1) % V.P),
                   lambda p: p["evals"]["z"][0].__setitem__(1, (p["evals"]["z"][0][1] + 1) % V.P)):
        bad = copy.deepcopy(proof)
        mutate(bad)
        with pytest.raises(V.Reject):
            V.verify(bad, air.program(), *tables, expect, bn_tables)


def test_bn128_transcripts_agree_and_are_order_sensitive(bn_tables):
    """the product's sponge and the checker's restatement of it, driven by the checker's permutation"""
    from eigen_zeth_amd.stark.transcript import TranscriptBN128
    from oracle import oracle as O
    O.p254_set(17, bn_tables[2], bn_tables[0], bn_tables[1])
    perm = lambda st: O.p254_perm([st], 17)[0]
    a, b, c = TranscriptBN128(perm), V.SpongeBN128(perm), TranscriptBN128(perm)
    for t in (a, b):
        t.absorb([1, 2, 3, 4, V.P - 1])
        t.absorb_root([123456789 << 100])
    c.absorb_root([123456789 << 100])
    c.absorb([1, 2, 3, 4, V.P - 1])
    x, y, z = a.squeeze(60), b.squeeze(60), c.squeeze(60)
    assert x == y and x != z and all(0 <= v < V.P for v in x) and len(set(x)) == 60


def test_grinding_is_refused_in_bn128_mode():
    with pytest.raises(AssertionError):
        PR.StarkParams(6, 1, 2, 3, 5, pow_bits=8, hash="bn128")


@pytest.mark.gpu
@pytest.mark.parametrize("name,logn,logb", [("chunk16", 8, 1), ("wide8", 10, 2)])
def test_bn128_gpu_proof_equals_cpu_proof(prover, tables, bn_tables, name, logn, logb):
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    air, tr, pub = _case(name, logn)
    params = PR.StarkParams(logn, logb, 3, 3, 6, hash="bn128")
    gpu = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover, hash_mode="bn128")))
    cpu = PR.proof_to_json(PR.prove(air, tr, pub, params, CpuBackend(*tables, hash_mode="bn128", bn_tables=bn_tables)))
    assert gpu == cpu
    assert V.verify(json.loads(gpu), air.program(), *tables, V.expectation(params.to_dict()), bn_tables)


@pytest.mark.gpu
def test_bn128_gpu_proof_2_16_verifies(prover, tables, bn_tables):
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    air, tr, pub = _case("chunk16", 16)
    params = PR.StarkParams(16, 2, 3, 5, 12, hash="bn128")
    proof = json.loads(PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover, hash_mode="bn128", quotient="program"))))
    assert V.verify(proof, air.program(), *tables, V.expectation(params.to_dict()), bn_tables)


@pytest.mark.gpu
@pytest.mark.parametrize("name,logn,logb,nq", [("fib", 6, 1, 5), ("chunk16", 8, 1, 6), ("wide8", 10, 2, 50), ("chunk16", 12, 2, 20)])
def test_bn128_one_call_prover_writes_the_same_proof(prover, tables, bn_tables, name, logn, logb, nq):
    """zp_stark_prove_bn128 against the Python orchestration over the same library, and the independent verifier"""
    from eigen_zeth_amd.stark.backend_hip import HipBackend
    air, tr, pub = _case(name, logn)
    params = PR.StarkParams(logn, logb, 3, 3, nq, hash="bn128")
    ref = PR.proof_to_json(PR.prove(air, tr, pub, params, HipBackend(prover=prover, hash_mode="bn128", quotient="program")))
    d_tr = prover.upload(tr)
    got = prover.stark_prove_bn128(air.name, air.program(), d_tr, [int(v) for v in pub], logn, logb, 3, 3, nq)
    d_tr.free()
    assert got == ref
    assert V.verify(json.loads(got), air.program(), *tables, V.expectation(params.to_dict()), bn_tables)


@pytest.mark.gpu
def test_bn254_sponge_in_one_call_equals_permutation_by_permutation(prover, bn_tables):
    import random
    from oracle import oracle as O
    prover.install_poseidon_bn254(17)
    O.p254_set(17, bn_tables[2], bn_tables[0], bn_tables[1])
    rnd = random.Random(4)
    for nblocks, extra in ((0, 0), (0, 2), (1, 0), (3, 1)):
        state = [rnd.randrange(V.R_BN254) for _ in range(17)]
        blocks = [[rnd.randrange(V.R_BN254) for _ in range(16)] for _ in range(nblocks)]
        st, rates = list(state), []
        if not blocks:
            st = O.p254_perm([st], 17)[0]
        for b in blocks:
            st = O.p254_perm([[st[0]] + b], 17)[0]
        rates.append(st[1:])
        for _ in range(extra):
            st = O.p254_perm([st], 17)[0]
            rates.append(st[1:])
        got_state, got_rates = prover.poseidon_bn254_sponge(state, blocks, extra)
        assert got_state == st and got_rates == rates
