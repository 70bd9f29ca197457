"""oracle/aggregate_verify.py -- the checker's verifier of an AGGREGATED proof (TEST INFRASTRUCTURE; imports nothing from the
product package).

An aggregated proof (what GenAggregatedProof returns: proto/prover/v1/prover.proto:115-126, consumed at
src/prover/provider.rs:436-451) holds
    "inner": every inner proof WITHOUT the authentication paths of its query openings (everything else: air name / digest,
             parameters, publics, roots, out-of-domain evaluations, FRI roots + final layer, grinding nonce, and per query the
             index and the opened values of every tree)
    "stark": a STARK over the Merkle-verifier AIR (the statement arrives as a constraint program blob, like every other AIR)
             whose public inputs are the inner proofs' roots, the leaf index and the opened values of every (slot, proof, tree).
Accepting means BOTH INNER PROOFS VERIFY, the work split in two:
  1. arithmetic, natively: every inner proof passes the whole verifier on the opened values as given -- parameters are the
     verifier's, the Fiat-Shamir transcript is replayed (which dictates the query indices), the constraint identity holds at the
     out-of-domain point, the DEEP quotient and every FRI fold are consistent at every query, the final layer is low degree, the
     grinding nonce is valid (stark_verify.verify(trust_openings=True): everything but the Merkle paths);
  2. the public inputs of the outer STARK are exactly those roots, indices and opened values (slot g re-opens query g mod n_queries);
  3. hashing, in the circuit: the outer STARK verifies under the verifier-AIR program -- for every slot / proof / tree the public
     values, hashed as a leaf and up a path along the bits of the public index, give the public root.
PARITY UNPINNED w.r.t. the external prover."""
from . import stark_verify as V
from .air_program import Program


def tree_depths(logn, logb, W2, sched):
    """leaf-index widths (= tree depths) of the committed trees in the order the verifier AIR lists them"""
    logm = logn + logb
    d = [logm] + ([logm] if W2 else []) + [logm]
    return d + [lg - f for (lg, f) in sched]


def roots_of(proof, W2):
    r = [proof["roots"]["trace"]] + ([proof["roots"]["stage2"]] if W2 else []) + [proof["roots"]["quotient"]]
    return r + list(proof["fri"]["roots"])


def verify(agg, inner_program, outer_program, rc, mds, inner_expect, outer_expect, n_slots):
    """agg: the aggregated proof (dict); inner_program / outer_program: constraint program blobs of the inner AIR and of the
    Merkle-verifier AIR; *_expect: the verifier's own STARK parameters for the two levels; n_slots: query slots of the outer
    trace (a property of the verifier AIR's layout, given by whoever supplies its program)."""
    inner = agg["inner"]
    if not inner:
        raise V.Reject("no inner proofs")
    heads = [V.verify(h, inner_program, rc, mds, inner_expect, trust_openings=True) for h in inner]
    W2 = heads[0]["W2"]
    depths = tree_depths(inner_expect["logn"], inner_expect["logb"], W2, heads[0]["sched"])
    want = []
    for h in inner:
        roots = roots_of(h, W2)
        if len(roots) != len(depths):
            raise V.Reject("inner proof has the wrong number of commitments")
        for r in roots:
            want += [int(v) for v in r]
    nq = inner_expect["n_queries"]
    for g in range(n_slots):
        for hd in heads:
            j = hd["indices"][g % nq]
            want += [j & ((1 << d) - 1) for d in depths]
    for g in range(n_slots):
        for h in inner:
            q = h["queries"][g % nq]
            for part in [q["trace"]] + ([q["stage2"]] if W2 else []) + [q["quotient"]] + list(q["fri"]):
                want += [int(v) for v in part["values"]]
    outer = agg["stark"]
    if [int(v) for v in outer["publics"]] != want:
        raise V.Reject("the outer proof's public inputs are not the inner proofs' roots, query indices and opened values")
    prog = outer_program if isinstance(outer_program, Program) else Program(outer_program)
    return V.verify(outer, prog, rc, mds, outer_expect)
