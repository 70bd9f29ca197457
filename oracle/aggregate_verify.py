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
     verifier's, the constraint identity holds at the out-of-domain point, the DEEP quotient and every FRI fold are consistent at
     every query, the final layer is low degree (stark_verify.verify(trust_openings=True): everything but the Merkle paths) --
     with the Fiat-Shamir transcript READ, not hashed: the outer public inputs list, in protocol order, what every permutation
     of the sponge absorbs and the rates the protocol reads (PublicSponge); the checker compares the absorbed blocks with the
     proof's data and takes challenges, grinding digest and query indices from the public rates;
  2. the public inputs of the outer STARK are exactly those roots, indices and opened values (slot g re-opens query g mod n_queries);
  3. hashing, in the circuit: the outer STARK verifies under the verifier-AIR program -- for every slot / proof / tree the public
     values, hashed as a leaf and up a path along the bits of the public index, give the public root; the sponge permutations of
     every transcript map the public blocks to the public rates (chained through the capacity), the grinding hash maps seed and
     nonce to the public digest.  (The one hash left to the checker is the digest of the OUTER proof's own public inputs.)
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


def verify_tree(agg, leaf_program, leaf_expect, rc, mds, program_of_shape, expect_of_shape, bn_tables=None):
    """an aggregated proof of any level.  Level 1 aggregates chunk proofs (verify below).  Level n + 1 aggregates two aggregated
    proofs of level <= n: its "inner" are THEIR aggregation STARKs without paths, its "children" what those aggregated.  Accepting
    means every chunk proof at the leaves verifies: the outer STARK + native arithmetic vouch for the two aggregation STARKs one
    level down (verify), and each of those, now known to verify, vouches for the hashing of ITS inner proofs, whose arithmetic is
    checked natively against its public inputs (check_inner) -- and so on down to the chunk proofs.
    program_of_shape(shape dict) -> (program blob of the verifier AIR for inner proofs of that shape, its query slots);
    expect_of_shape(shape dict) -> the verifier's STARK parameters of a proof over that AIR."""
    def node_ok(node, stark_publics):
        """node = {"shape", "inner", ["children"]}: inner proofs vouched for by a STARK whose public inputs are stark_publics"""
        kids = node.get("children")
        if not kids:
            check_inner(node["inner"], stark_publics, leaf_program, rc, mds, leaf_expect, program_of_shape(node["shape"])[1])
            return
        if len(kids) != len(node["inner"]) or len({repr(sorted(k["shape"].items())) for k in kids}) != 1:
            raise V.Reject("children do not match the inner proofs")
        prog_below, _ = program_of_shape(kids[0]["shape"])
        check_inner(node["inner"], stark_publics, prog_below, rc, mds, expect_of_shape(kids[0]["shape"]), program_of_shape(node["shape"])[1])
        for kid, hdr in zip(kids, node["inner"]):
            node_ok(kid, hdr["publics"])            # hdr (an aggregation STARK one level down) verifies: its publics are facts
    top_prog, _ = program_of_shape(agg["shape"])
    node_ok(agg, agg["stark"]["publics"])
    prog = top_prog if isinstance(top_prog, Program) else Program(top_prog)
    return V.verify(agg["stark"], prog, rc, mds, expect_of_shape(agg["shape"]), bn_tables)


def check_inner(inner, outer_publics, inner_program, rc, mds, inner_expect, n_slots):
    """steps 1 and 2 of verify(): the inner proofs (without paths) pass the native part of the verifier with their transcripts
    read off `outer_publics`, and `outer_publics` are exactly their roots, indices, opened values and transcripts.  Returns
    nothing; whoever calls it must have a reason to believe a STARK with these public inputs verifies."""
    _check_inner(inner, [int(v) for v in outer_publics], inner_program, rc, mds, inner_expect, n_slots)


def verify(agg, inner_program, outer_program, rc, mds, inner_expect, outer_expect, n_slots, bn_tables=None):
    """agg: the aggregated proof (dict); inner_program / outer_program: constraint program blobs of the inner AIR and of the
    Merkle-verifier AIR; *_expect: the verifier's own STARK parameters for the two levels; n_slots: query slots of the outer
    trace (a property of the verifier AIR's layout, given by whoever supplies its program).  bn_tables: the Poseidon-BN254
    tables when the OUTER proof is in BN128-hash mode (the final STARK over an aggregated proof's STARK: the inner proofs are
    Goldilocks-mode either way)."""
    outer = agg["stark"]
    _check_inner(agg["inner"], [int(v) for v in outer["publics"]], inner_program, rc, mds, inner_expect, n_slots)
    prog = outer_program if isinstance(outer_program, Program) else Program(outer_program)
    return V.verify(outer, prog, rc, mds, outer_expect, bn_tables)


def _check_inner(inner, outer_publics, inner_program, rc, mds, inner_expect, n_slots):
    if not inner:
        raise V.Reject("no inner proofs")
    # the Merkle part of the outer public inputs has a length the shapes fix; what follows it is the inner transcripts, proof by
    # proof: absorbed blocks and read rates in protocol order.  Every inner proof is verified on a sponge that READS them.
    probe = Program(inner_program) if not isinstance(inner_program, Program) else inner_program
    sched, _ = V.fri_schedule(inner_expect["logn"], inner_expect["logb"], inner_expect["fri_logf"], inner_expect["fri_final_log"])
    T = 2 + (1 if probe.width2 else 0) + len(sched)
    per_query = probe.width + probe.width2 + 3 * probe.q_chunks + sum(3 << f for (_, f) in sched)
    n_merkle = len(inner) * T * 4 + n_slots * len(inner) * (T + per_query)
    sponge = V.PublicSponge(outer_publics[n_merkle:])
    heads = []
    for h in inner:
        sponge.queue, sponge.avail = [], []          # a fresh sponge per proof on the one stream
        heads.append(V.verify(h, probe, rc, mds, inner_expect, trust_openings=True, public_transcript=sponge))
    if sponge.pos != len(sponge.stream):
        raise V.Reject("the outer proof's public inputs hold more transcript than the inner proofs have")
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
    if outer_publics[:n_merkle] != want:
        raise V.Reject("the outer proof's public inputs are not the inner proofs' roots, query indices and opened values")
