"""oracle/wrap_verify.py -- the checker's side of the Groth16 wrap (TEST INFRASTRUCTURE; imports nothing from the product package).

The wrap's circuit (product: eigen_zeth_amd/service/wrap_circuit.py) has ONE public input d and proves: "there are leaf elements and
authentication paths such that the leaves hash up to the roots AT THE INDICES THE TRANSCRIPT DICTATES -- the sponge over the statement's
parameters and digest, the public inputs (or their commitment), the roots, the out-of-domain evaluations and the final layer, hashed inside the
circuit (stage B-1, round 5) -- and d is the root of a 16-ary Poseidon-BN254 tree over the list  roots | aux | transcript data | per query:
index, the leaf elements of every tree".  A Groth16 proof with public input d is therefore worth exactly as much as one's knowledge of what d
commits to: `public_input` recomputes d from a final STARK's data by this module's own packing -- whoever holds the final STARK checks d, then
verifies THE ARITHMETIC of the STARK (out-of-domain identity, DEEP, folds: `verify_rest`) on the opened values as given, without their
authentication paths and WITHOUT RE-DERIVING THE QUERY INDICES: both are the circuit's business now.  (Round 4's circuit left the indices
free, so its checker had to hash the transcript again and compare.)  `verify` = the comparison of d + the pairing check
(oracle/groth16_verify.py).  PARITY UNPINNED w.r.t. the external prover."""
from . import groth16_verify as GV
from . import naive as NV
from . import stark_verify as V


def leaf_elements(values):
    """the field elements a leaf's sponge absorbs: blocks of 56 values, 16 elements each (naive.pack_leaf_block)"""
    out = []
    for off in range(0, max(len(values), 1), NV.LEAF_BLOCK):
        out += NV.pack_leaf_block(values[off:off + NV.LEAF_BLOCK])
    return out


def _pack3(vals):
    P = V.P
    v = [int(x) % P for x in vals] + [0] * (-len(vals) % 3)
    return [v[i] + (v[i + 1] << 64) + (v[i + 2] << 128) for i in range(0, len(v), 3)]


def transcript_data(final_stark, program, bn_tables):
    """everything the final STARK's sponge absorbs besides the Merkle roots, as the field elements it absorbs them in, in order: the parameters,
    domain and statement digest (with the public inputs when they are few, else followed by their commitment), one element per out-of-domain
    evaluation, the final layer plane by plane (oracle/stark_verify.py states the same order for the verifier's own sponge)"""
    from .air_program import Program
    air = program if isinstance(program, Program) else Program(program)
    pr, pubs = final_stark["params"], [int(v) for v in final_stark["publics"]]
    head = [pr["logn"], pr["logb"], air.width, air.width2, pr["fri_logf"], pr["fri_final_log"], pr["n_queries"], pr["pow_bits"], int(final_stark["root32"]),
            int(final_stark["shift"])] + air.digest_words() + [len(pubs)]
    if len(pubs) <= V.PUBLICS_INLINE:
        out = _pack3(head + pubs)
    else:
        V.O.p254_set(17, bn_tables[2], bn_tables[0], bn_tables[1])
        out = _pack3(head) + [int(V.publics_digest(pubs, None, None, True)[0])]
    out += [_pack3(r)[0] for r in final_stark["evals"]["z"] + final_stark["evals"]["zw"]]
    for plane in final_stark["fri"]["final"]:
        out += _pack3(plane)
    return out


def public_input(final_stark, aux, bn_tables, program):
    """d for a final STARK in BN128-hash mode; bn_tables = (rc, mds, rp) of the width-17 instance; program: the statement (its digest is
    transcript data)"""
    rc, mds, rp = bn_tables
    roots = [final_stark["roots"]["trace"], final_stark["roots"]["quotient"]] + list(final_stark["fri"]["roots"])
    data = [int(r[0]) for r in roots] + [int(aux) % NV.BN254_R] + transcript_data(final_stark, program, bn_tables)
    for q in final_stark["queries"]:
        data.append(int(q["index"]))
        for part in [q["trace"], q["quotient"]] + list(q["fri"]):
            data += leaf_elements([int(v) for v in part["values"]])
    lvl = data
    while len(lvl) > 1:
        lvl = lvl + [0] * (-len(lvl) % 16)
        lvl = [NV.poseidon_bn254_perm([0] + lvl[i:i + 16], rc, mds, rp)[0] for i in range(0, len(lvl), 16)]
    return lvl[0]


def verify_rest(final_stark, program, rc, mds, expect, bn_tables):
    """what a holder of (final STARK, Groth16 proof with the right d) still checks natively: parameters, the out-of-domain constraint identity, the
    DEEP quotient and every fold at every query -- on the opened values and AT THE INDICES as given (no authentication path is read, no index is
    compared with the transcript: the circuit proved both), the challenges read off the verifier's own sponge.  Raises V.Reject."""
    return V.verify(final_stark, program, rc, mds, expect, bn_tables, trust_openings=True, trust_indices=True)


def verify(vk, proof, pub, final_stark, aux, bn_tables, program):
    """proof: {"pi_a", "pi_b", "pi_c"} points; pub: [d]; vk: the verifying key's points"""
    if len(pub) != 1 or int(pub[0]) != public_input(final_stark, aux, bn_tables, program):
        raise V.Reject("the public input is not the commitment to this final STARK's roots, indices and openings")
    if not GV.verify(vk, proof, pub):
        raise V.Reject("the Groth16 proof does not verify")
    return True
