"""oracle/aggregate_verify.py -- the checker's verifier of an AGGREGATED proof (TEST INFRASTRUCTURE; imports nothing from the
product package).

An aggregated proof (what GenAggregatedProof returns: proto/prover/v1/prover.proto:115-126, consumed at
src/prover/provider.rs:436-451) holds
    "inner": the HEADER of every inner proof -- air name / digest, parameters, publics, roots, out-of-domain evaluations, FRI roots + final
             layer, grinding nonce.  NO query openings (round 4: the opened values are private witnesses of the verifier AIR)
    "stark": a STARK over the verifier AIR (the statement arrives as a constraint program blob, like every other AIR) whose public inputs
             are the inner proofs' roots, the leaf index of every (slot, proof, tree), their transcripts, and the constants of the
             arithmetic constraints (arithmetic_publics) + the final-layer value at every query's position.
Accepting means BOTH INNER PROOFS VERIFY, the work split in two:
  1. per inner proof, natively, what needs no opening: parameters are the verifier's, the constraint identity holds at the out-of-domain
     point, the final layer is low degree, grinding (stark_verify.verify(header_only=True)) -- with the Fiat-Shamir transcript READ, not
     hashed: the outer public inputs list, in protocol order, what every permutation of the sponge absorbs and the rates the protocol
     reads (PublicSponge); the checker compares the absorbed blocks with the header's data and takes challenges, grinding digest and
     query indices from the public rates;
  2. the public inputs of the outer STARK are exactly what the headers dictate (roots, indices, transcripts, arithmetic constants, final-layer
     values: slot g re-opens query g mod n_queries);
  3. everything at the queries, in the circuit: the outer STARK verifies under the verifier-AIR program -- for every slot / proof / tree
     there are values that hash, as a leaf and up a path along the bits of the public index, to the public root; those values give the
     DEEP quotient at the query point, every layer's opened coset interpolates the value the previous layer claims and folds to the next
     one's, the last fold is the public final-layer value; the sponge permutations of every transcript map the public blocks to the
     public rates, the grinding hash maps seed and nonce to the public digest.  (The one hash left to the checker is the digest of the OUTER
     proof's own public inputs.)
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


def _e3_pow_list(base, n):
    """[base^1 .. base^n] in F_p^3"""
    out, cur = [], [1, 0, 0]
    for _ in range(n):
        cur = V.NV.e3_mul(cur, base)
        out.append(cur)
    return out


def arithmetic_publics(h, head, expect):
    """What the verifier AIR takes as PUBLIC per inner proof for its arithmetic constraints -- everything a verifier derives from the proof's
    header alone (the checker's own restatement; the product computes the same list in stark/verifier_air.py:arith_publics):
    g^1..g^8 with g = 1/gamma; gamma^(Wall-1); gamma^(Wall+Wt-1); E_z = sum_k gamma^k ev_k(zeta); E_zw = sum_k gamma^(Wall+k) ev_k(zeta w); zeta; zeta w;
    beta_l^j (1 <= j < 2^f) for every committed FRI layer.  O(columns) field operations."""
    NV = V.NV
    gamma, zeta = [int(v) for v in head["gamma"]], [int(v) for v in head["zeta"]]
    out = [v for e in _e3_pow_list(NV.e3_inv(gamma), 8) for v in e]
    Wt, Wall = head["W"] + head["W2"], head["W"] + head["W2"] + head["Wq"]
    gp = [[1, 0, 0]] + _e3_pow_list(gamma, Wall + Wt - 1)
    eza, ezb = [0, 0, 0], [0, 0, 0]
    for k in range(Wall):
        eza = NV.e3_add(eza, NV.e3_mul(gp[k], [int(v) for v in h["evals"]["z"][k]]))
    for k in range(Wt):
        ezb = NV.e3_add(ezb, NV.e3_mul(gp[Wall + k], [int(v) for v in h["evals"]["zw"][k]]))
    wN = NV.root(expect["logn"], expect["root32"])
    out += gp[Wall - 1] + gp[Wall + Wt - 1] + eza + ezb + zeta + [v * wN % V.P for v in zeta]
    for (_, f), beta in zip(head["sched"], head["betas"]):
        out += [v for e in _e3_pow_list([int(v) for v in beta], (1 << f) - 1) for v in e]
    return out


def _check_inner(inner, outer_publics, inner_program, rc, mds, inner_expect, n_slots):
    """The inner proofs arrive as HEADERS: parameters, publics, roots, out-of-domain evaluations, FRI roots, final layer, grinding nonce -- no
    query openings at all.  The verifier-AIR STARK vouches for everything that happens at the queries (paths, DEEP quotient, folds); what is
    checked here, natively, is the part of a verifier that needs no opening (stark_verify.verify(header_only=True): parameters, the transcript
    READ off the public inputs, the out-of-domain constraint identity, the low-degree test of the final layer, grinding) and that the outer
    public inputs are what the headers dictate: roots; the leaf index of every (slot, proof, tree), from the transcript's query indices; the
    transcripts; the arithmetic constants (arithmetic_publics); the final-layer value at every query's position (a lookup)."""
    if not inner:
        raise V.Reject("no inner proofs")
    probe = Program(inner_program) if not isinstance(inner_program, Program) else inner_program
    sched, final_log = V.fri_schedule(inner_expect["logn"], inner_expect["logb"], inner_expect["fri_logf"], inner_expect["fri_final_log"])
    if not sched:
        raise V.Reject("inner proofs without a committed FRI layer are not aggregated")
    T = 2 + (1 if probe.width2 else 0) + len(sched)
    n_merkle = len(inner) * T * 4 + n_slots * len(inner) * T
    for h in inner:
        if "queries" in h:
            raise V.Reject("an aggregated proof carries no openings of its inner proofs")
    # first pass: lengths (the transcript section is delimited by what follows it, whose length the shapes fix)
    n_arith = 42 + sum(3 * ((1 << f) - 1) for (_, f) in sched)
    n_tail = len(inner) * n_arith + n_slots * len(inner) * 3
    if len(outer_publics) < n_merkle + n_tail:
        raise V.Reject("the outer proof has too few public inputs")
    sponge = V.PublicSponge(outer_publics[n_merkle:len(outer_publics) - n_tail])
    heads = []
    for h in inner:
        sponge.queue, sponge.avail = [], []          # a fresh sponge per proof on the one stream
        heads.append(V.verify(h, probe, rc, mds, inner_expect, header_only=True, public_transcript=sponge))
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
    if outer_publics[:n_merkle] != want:
        raise V.Reject("the outer proof's public inputs are not the inner proofs' roots and query indices")
    tail = []
    for h, hd in zip(inner, heads):
        tail += arithmetic_publics(h, hd, inner_expect)
    mask = (1 << final_log) - 1
    for g in range(n_slots):
        for h, hd in zip(inner, heads):
            pos = hd["indices"][g % nq] & mask
            tail += [int(h["fri"]["final"][c][pos]) for c in range(3)]
    if outer_publics[len(outer_publics) - n_tail:] != tail:
        raise V.Reject("the outer proof's arithmetic public inputs are not what the inner proofs' headers dictate")
