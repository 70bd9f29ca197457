"""oracle/wrap_verify.py -- the checker's side of the Groth16 wrap (TEST INFRASTRUCTURE; imports nothing from the product package).

The wrap's circuit (product: eigen_zeth_amd/service/wrap_circuit.py) has ONE public input d and proves: "there are query indices, leaf elements
and authentication paths such that the leaves hash up to the roots at those indices, and d is the root of a 16-ary Poseidon-BN254 tree over
the list  roots | aux | per query: index, the leaf elements of every tree".  A Groth16 proof with public input d is therefore worth exactly
as much as one's knowledge of what d commits to: `public_input` recomputes d from a final STARK (its roots, indices and opened VALUES, packed
into field elements by this module's own pack_leaf_block) -- whoever holds the final STARK checks d, then verifies the rest of the STARK
(transcript, out-of-domain identity, DEEP, folds: oracle/stark_verify.py) WITHOUT its authentication paths.  `verify` = that comparison +
the pairing check (oracle/groth16_verify.py).  PARITY UNPINNED w.r.t. the external prover."""
from . import groth16_verify as GV
from . import naive as NV
from . import stark_verify as V


def leaf_elements(values):
    """the field elements a leaf's sponge absorbs: blocks of 56 values, 16 elements each (naive.pack_leaf_block)"""
    out = []
    for off in range(0, max(len(values), 1), NV.LEAF_BLOCK):
        out += NV.pack_leaf_block(values[off:off + NV.LEAF_BLOCK])
    return out


def public_input(final_stark, aux, bn_tables):
    """d for a final STARK in BN128-hash mode; bn_tables = (rc, mds, rp) of the width-17 instance"""
    rc, mds, rp = bn_tables
    roots = [final_stark["roots"]["trace"], final_stark["roots"]["quotient"]] + list(final_stark["fri"]["roots"])
    data = [int(r[0]) for r in roots] + [int(aux) % NV.BN254_R]
    for q in final_stark["queries"]:
        data.append(int(q["index"]))
        for part in [q["trace"], q["quotient"]] + list(q["fri"]):
            data += leaf_elements([int(v) for v in part["values"]])
    lvl = data
    while len(lvl) > 1:
        lvl = lvl + [0] * (-len(lvl) % 16)
        lvl = [NV.poseidon_bn254_perm([0] + lvl[i:i + 16], rc, mds, rp)[0] for i in range(0, len(lvl), 16)]
    return lvl[0]


def verify(vk, proof, pub, final_stark, aux, bn_tables):
    """proof: {"pi_a", "pi_b", "pi_c"} points; pub: [d]; vk: the verifying key's points"""
    if len(pub) != 1 or int(pub[0]) != public_input(final_stark, aux, bn_tables):
        raise V.Reject("the public input is not the commitment to this final STARK's roots, indices and openings")
    if not GV.verify(vk, proof, pub):
        raise V.Reject("the Groth16 proof does not verify")
    return True
