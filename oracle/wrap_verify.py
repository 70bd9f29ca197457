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
(oracle/groth16_verify.py).

Round 6, stage B-2: the circuit ALSO runs the verifier's field arithmetic (service/wrap_arith.py: challenges off the in-circuit sponge, the constraint
identity at zeta, DEEP, every fold, the final layer's degree), so its public input commits to PUBLIC data only -- `public_input_b2`: aux | the
parameter / digest elements (when the public inputs are few and sit among them) | the commitment to the public inputs | zeta | the statement's sparse
fixed columns at zeta.  No part of the final STARK is needed to check a proof: `verify_b2` recomputes the fixed columns at the zeta the proof text
names -- a function of (statement, public inputs, zeta) -- recomputes d, and runs the pairing check.  verify_rest is not part of that path any more;
it stays for the hashing-only circuit of rounds 4-5 (which the tests still build).  PARITY UNPINNED w.r.t. the external prover."""
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


def fixed_columns_at(program, publics, zeta, logn, root32):
    """the statement's SPARSE fixed columns (columns 2.. of the program) at the out-of-domain point, by the checker's own reader of the program blob"""
    from .air_program import Program
    air = program if isinstance(program, Program) else Program(program)
    return [air.fixed_eval_ext(k, [int(v) for v in publics], [int(v) % V.P for v in zeta], logn, root32) for k in range(len(air.fixed_cols))]


def public_input_b2(program, params, root32, shift, publics, aux, zeta_words, bn_tables):
    """d of a stage B-2 wrap from PUBLIC data: the statement (constraint program + the STARK parameters and domain it is proven under), its public
    inputs, the element the proof is bound to, and zeta_words -- the three 64-bit words of the challenge as the proof text carries them (a word may
    be >= p: it is then a second name of its residue; the commitment holds the words, the evaluation takes them mod p)."""
    from .air_program import Program
    rc, mds, rp = bn_tables
    air = program if isinstance(program, Program) else Program(program)
    pubs = [int(v) for v in publics]
    head = [params["logn"], params["logb"], air.width, air.width2, params["fri_logf"], params["fri_final_log"], params["n_queries"], params.get("pow_bits", 0),
            int(root32), int(shift)] + air.digest_words() + [len(pubs)]
    data = [int(aux) % NV.BN254_R]
    if len(pubs) <= V.PUBLICS_INLINE:
        data += _pack3(head + pubs)
    else:
        V.O.p254_set(17, rp, rc, mds)
        data += [int(V.publics_digest(pubs, None, None, True)[0])]
    zw = [int(v) for v in zeta_words]
    if len(zw) != 3 or any(not 0 <= v < (1 << 64) for v in zw):
        raise V.Reject("malformed zeta")
    data.append(zw[0] + (zw[1] << 64) + (zw[2] << 128))
    for fz in fixed_columns_at(air, pubs, zw, params["logn"], root32):
        data.append(int(fz[0]) + (int(fz[1]) << 64) + (int(fz[2]) << 128))
    lvl = data
    while len(lvl) > 1:
        lvl = lvl + [0] * (-len(lvl) % 16)
        lvl = [NV.poseidon_bn254_perm([0] + lvl[i:i + 16], rc, mds, rp)[0] for i in range(0, len(lvl), 16)]
    return lvl[0]


def verify_b2(vk, proof, pub, program, params, root32, shift, publics, aux, zeta_words, bn_tables):
    """everything a reader of a stage B-2 final proof does: d from public data, the pairing check.  (vk must be the key of the circuit built for this
    statement: the circuit pins the transcript's parameter block to it.)"""
    if len(pub) != 1 or int(pub[0]) != public_input_b2(program, params, root32, shift, publics, aux, zeta_words, bn_tables):
        raise V.Reject("the public input is not the commitment to this statement, these public inputs and this zeta")
    if not GV.verify(vk, proof, pub):
        raise V.Reject("the Groth16 proof does not verify")
    return True
