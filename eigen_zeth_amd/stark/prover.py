"""Chunk STARK prover: trace -> LDE -> commit -> constraint quotient -> commit -> out-of-domain
evaluations -> DEEP quotient -> FRI -> query openings.  Backend-agnostic orchestration; every
O(trace) step is one backend call (GPU backend: backend_hip.py, one C-ABI call each).

Serves GenChunkProof (proto/prover/v1/prover.proto:56-66; client src/prover/provider.rs:358-390).
The protocol follows the public DEEP-FRI STARK construction; it is NOT
claimed bit-compatible with the external eigen-zkvm prover (parity unpinned)."""
from __future__ import annotations

import json
import time

from . import air as air_mod
from . import field as F
from .transcript import Transcript, TranscriptBN128

P = F.P
PUBLICS_INLINE = 64     # public inputs beyond this many are absorbed through their Merkle digest (publics_rows)


def publics_rows(pubs, bn):
    """the matrix a long public-input vector is committed as: Goldilocks mode: rows of 8 values, row count a power of two
    (>= 2), row-major [M][8];  BN128 mode: rows of 48 values (16 field elements), column-major [48][M].  Zero padded."""
    import numpy as np
    v = np.array([int(x) % P for x in pubs], dtype=np.uint64)
    if not bn:
        M = 2
        while M * 8 < len(v):
            M <<= 1
        out = np.zeros((M, 8), dtype=np.uint64)
        out.reshape(-1)[:len(v)] = v
        return out
    M = max(1, -(-len(v) // 48))
    rows = np.zeros((M, 48), dtype=np.uint64)
    rows.reshape(-1)[:len(v)] = v
    return np.ascontiguousarray(rows.T)


def bn128_rows_per_leaf_log(width, logm):
    """BN128-hash mode: a leaf of a committed tree (trace, stage 2, quotient) holds 2^g rows i, i + M', i + 2M', ... (M' = M / 2^g; the
    grouping of a FRI layer: a free reinterpretation of the column-major matrix as [width 2^g][M']), g the largest with
    width * 2^g <= 56 values = ONE width-17 permutation per leaf (56 values to a sponge block: oracle/naive.py pack_leaf_block).  The
    verifier AIR's 26 columns: 2 rows per leaf; its 9-column quotient: 4 -- instead of a permutation per row.  The verifier takes
    row j from position j / M' of leaf j mod M'.  (Goldilocks mode: 1 row per leaf -- the verifier AIR's schedule is written for that.)"""
    g = 0
    while (width << (g + 1)) <= 56 and g + 1 <= logm - 4:
        g += 1
    return g


class StarkParams:
    """Security (conjectured, ethSTARK-style): n_queries * logb + pow_bits bits -- each query of a rate-2^-logb code
    rejects a far word with probability 1 - 2^-logb, and the prover must grind pow_bits of Poseidon work before it
    learns the query positions.  The service default (engine.EngineConfig) is 80 queries, blow-up 2, 20 bits = 100."""

    def __init__(self, logn, logb=1, fri_logf=3, fri_final_log=6, n_queries=24, pow_bits=0, hash="gl"):
        """hash: "gl" = Poseidon over Goldilocks, binary Merkle trees (chunk / recursion STARKs);  "bn128" = Poseidon over the
        BN254 scalar field, 16-ary trees, transcript over the same field (the last STARK before the Groth16 wrap)."""
        assert hash in ("gl", "bn128")
        assert hash == "gl" or pow_bits == 0, "the grinding hash is Goldilocks-Poseidon: BN128 mode takes its bits from queries"
        self.logn, self.logb = logn, logb
        self.fri_logf, self.fri_final_log, self.n_queries, self.pow_bits = fri_logf, fri_final_log, n_queries, pow_bits
        self.hash = hash

    def to_dict(self):
        d = dict(logn=self.logn, logb=self.logb, fri_logf=self.fri_logf, fri_final_log=self.fri_final_log,
                 n_queries=self.n_queries, pow_bits=self.pow_bits)
        if self.hash != "gl":
            d["hash"] = self.hash
        return d

    @staticmethod
    def from_dict(d):
        return StarkParams(d["logn"], d["logb"], d["fri_logf"], d["fri_final_log"], d["n_queries"], d.get("pow_bits", 0),
                           d.get("hash", "gl"))

    def security_bits(self):
        return self.n_queries * self.logb + self.pow_bits

    def fri_schedule(self):
        """list of (log size of the layer that is committed and folded, log fold factor)"""
        logm = self.logn + self.logb
        sched = []
        cur = logm
        while cur > self.fri_final_log + self.logb:
            f = min(self.fri_logf, cur - (self.fri_final_log + self.logb))
            sched.append((cur, f))
            cur -= f
        return sched, cur


def _ints(a):
    if hasattr(a, "tolist"):      # numpy array (any rank): converted by a C loop into (nested) lists of Python ints
        return a.tolist()
    return [int(v) for v in a]


def prove(air, trace, pubs, params, be, timings=None):
    """trace: uint64 [W][N] host array (the witness), pubs: public inputs; be: backend."""
    t_all = time.perf_counter()
    tm = {} if timings is None else timings

    def tick(name, t0):
        be.sync()
        tm[name] = tm.get(name, 0.0) + (time.perf_counter() - t0)

    logn, logb = params.logn, params.logb
    logm = logn + logb
    N, M, W = 1 << logn, 1 << logm, air.width
    assert tuple(trace.shape) == (W, N)
    shift, root32 = be.shift, be.root32
    wN = F.root(logn, root32)
    W2 = air.width2
    assert len(pubs) == air.n_pub
    # every parameter the verifier relies on is bound into the transcript (a proof cannot choose its own security level)
    bn = params.hash == "bn128"
    assert getattr(be, "hash_mode", "gl") == params.hash, "backend and parameters disagree on the hash mode"
    tr = TranscriptBN128(be.poseidon_bn254_perm17) if bn else Transcript(be.poseidon_perm, getattr(be, "poseidon_sponge", None))
    head = [logn, logb, W, W2, params.fri_logf, params.fri_final_log, params.n_queries, params.pow_bits, int(root32), int(shift)] \
        + air.digest_words() + [len(pubs)]
    if len(pubs) <= PUBLICS_INLINE:
        tr.absorb(head + _ints(pubs))
    else:
        # a long public-input vector (a verifier AIR names every root, index and opened value of its inner proofs: tens of
        # thousands of values) enters the transcript as ONE commitment: a sequential sponge over it would be thousands of
        # dependent permutations, the tree below is one parallel pass (rows of 8 / of 48 values, zero padded)
        tr.absorb(head)
        tr.absorb_root(be.publics_digest(_ints(pubs)))
    root_out = (lambda r: [str(int(r[0]))]) if bn else _ints      # a BN128 root is one 254-bit field element

    # 1. commit the trace
    t0 = time.perf_counter()
    Wt = W + W2                      # committed base columns: trace, then the stage-2 columns
    gt = bn128_rows_per_leaf_log(W, logm) if bn else 0                    # BN128 mode: 2^g rows per leaf
    g2 = bn128_rows_per_leaf_log(W2, logm) if (bn and W2) else 0
    c1 = be.commit_trace(trace, logn, logb, W2, gt)
    tick("lde+merkle(trace)", t0)
    tr.absorb_root(c1.root)
    chal, c2 = [], None
    if air.stage2:
        # stage 2: a challenge that depends on the first commitment, then the grand-product column
        t0 = time.perf_counter()
        chal = tr.challenge_e3()
        c2 = be.commit_stage2(air, c1, chal, logn, logb, g2)
        tick("grand-product+lde+merkle(stage2)", t0)
        tr.absorb_root(c2.root)
    alpha = tr.challenge_e3()

    # 2. constraint quotient on the coset, committed as 3 base columns
    t0 = time.perf_counter()
    fixed = be.fixed_ext(logn, logb, air, _ints(pubs)) if air.fixed_cols else be.fixed_ext(logn, logb)
    K = len(air.constraints)
    apow, cur = [], [1, 0, 0]
    for _ in range(K):
        apow.append(cur)
        cur = F.e3_mul(cur, alpha)
    sN = pow(shift, N, P)
    wb = F.root(logb, root32)
    zhinv = [F.inv((sN * pow(wb, j, P) - 1) % P) for j in range(1 << logb)]
    d_q = be.quotient(air, c1, fixed, list(_ints(pubs)) + list(chal), apow, zhinv, logn, logb, F.inv(wN))
    tick("quotient", t0)
    t0 = time.perf_counter()
    Q = air_mod.quotient_chunks(air)                 # pieces of degree < N the quotient is committed in
    assert Q <= (1 << logb), "the blow-up must cover the quotient degree: constraints of degree d need blow-up >= d - 1"
    if Q == 1:
        q_logn, Wq = logm, 3                         # committed as it stands: the quotient never leaves the evaluation form
    else:
        # q(x) = sum_j (x / shift)^(jN) qt_j(x), qt_j(shift X) = sum_{i<N} c'_(jN+i) X^i with c' the coefficients of q_c(shift X): the
        # pieces are slices of that vector; their LDEs (3Q base columns, piece-major) are what gets committed and opened
        d_q = be.quotient_pieces(d_q, logn, logb, Q)
        q_logn, Wq = logn, 3 * Q
    qg = bn128_rows_per_leaf_log(Wq, logm) if bn else 0       # BN128 mode: 2^qg rows of the quotient per leaf
    cq = be.commit_cols(d_q, M >> qg, Wq << qg)
    tick("merkle+intt(quotient)", t0)
    tr.absorb_root(cq.root)
    zeta = tr.challenge_e3()

    # 3. out-of-domain evaluations
    t0 = time.perf_counter()
    zeta_w = F.e3_scale(zeta, wN)
    # p_k(zeta), p_k(zeta w) for the Wt committed base columns, q-columns at zeta.  The backend decides how: the GPU backend reads
    # the resident extensions (barycentric form, zp_ood_eval: no coefficient buffer is kept since round 5), the CPU checker's
    # backend interpolates and evaluates the coefficient form (the definition)
    ev_z, ev_zw, ev_q = be.ood_evals(c1, Wt, d_q, Wq, q_logn, logn, logb, zeta, zeta_w)
    tick("ood-evals", t0)
    ev_all = [_ints(r) for r in ev_z] + [_ints(r) for r in ev_q]
    ev_next = [_ints(r) for r in ev_zw]
    for r in ev_all + ev_next:
        tr.absorb(r)
    gamma = tr.challenge_e3()

    # 4. DEEP quotient
    t0 = time.perf_counter()
    d_f = be.deep(c1.ext, Wt, d_q, Wq, logm, Wt, zeta, zeta_w, gamma, ev_all, ev_next)
    tick("deep", t0)

    # 5. FRI
    t0 = time.perf_counter()
    sched, final_log = params.fri_schedule()
    layers = []
    cur_shift = shift
    d_layer = d_f
    for (lg, f) in sched:
        com = be.commit_cols(d_layer, 1 << (lg - f), 3 << f)   # leaf = the 2^f * 3 values folded together
        tr.absorb_root(com.root)
        beta = tr.challenge_e3()
        d_next = be.fri_fold(d_layer, lg, f, beta, cur_shift)
        layers.append((lg, f, com, d_layer))
        d_layer = d_next
        cur_shift = pow(cur_shift, 1 << f, P)
    final = be.download(d_layer, (3, 1 << final_log))
    tick("fri", t0)
    final_l = [_ints(final[c]) for c in range(3)]
    for c in range(3):
        tr.absorb(final_l[c])

    # 6. proof of work, then queries
    t0 = time.perf_counter()
    pow_nonce = None
    if params.pow_bits:
        pow_nonce = int(be.pow_grind(tr.squeeze(4), params.pow_bits))
        tr.absorb([pow_nonce])
    qidx = tr.indices(params.n_queries, logm)
    t_rows = [j & ((M >> gt) - 1) for j in qidx]
    q_trace_vals = be.gather_rows(c1.ext, M >> gt, W << gt, t_rows)
    q_trace_paths = be.open_paths(c1.tree, M >> gt, t_rows)
    if c2 is not None:
        s_rows = [j & ((M >> g2) - 1) for j in qidx]
        q_s2_vals = be.gather_rows(be.column_view(c1.ext, W, M), M >> g2, W2 << g2, s_rows)
        q_s2_paths = be.open_paths(c2.tree, M >> g2, s_rows)
    q_rows = [j & ((M >> qg) - 1) for j in qidx]
    q_q_vals = be.gather_rows(d_q, M >> qg, Wq << qg, q_rows)
    q_q_paths = be.open_paths(cq.tree, M >> qg, q_rows)
    fri_open = []
    pos = list(qidx)
    for (lg, f, com, d_l) in layers:
        m = 1 << (lg - f)
        rows = [p & (m - 1) for p in pos]
        vals = be.gather_rows(d_l, m, 3 << f, rows)
        paths = be.open_paths(com.tree, m, rows)
        fri_open.append((rows, vals, paths))
        pos = rows
    tick("queries", t0)

    if bn:    # a BN128 path: per level the 16 digests of the group (decimal strings: 254-bit values)
        _path = lambda pth: [[str(int(v)) for v in lvl] for lvl in pth]
    else:
        _path = lambda pth: [_ints(x) for x in pth]
    queries = []
    for i, j in enumerate(qidx):
        queries.append({
            "index": int(j),
            "trace": {"values": _ints(q_trace_vals[i]), "path": _path(q_trace_paths[i])},
            "quotient": {"values": _ints(q_q_vals[i]), "path": _path(q_q_paths[i])},
            **({"stage2": {"values": _ints(q_s2_vals[i]), "path": _path(q_s2_paths[i])}} if c2 is not None else {}),
            "fri": [{"values": _ints(fo[1][i]), "path": _path(fo[2][i])} for fo in fri_open],
        })
    proof = {
        "air": air.name, "air_digest": air.digest(), "params": params.to_dict(),
        "root32": int(root32), "shift": int(shift),
        "publics": _ints(pubs),
        "roots": {"trace": root_out(c1.root), "quotient": root_out(cq.root), **({"stage2": root_out(c2.root)} if c2 is not None else {})},
        "evals": {"z": ev_all, "zw": ev_next},
        "fri": {"roots": [root_out(l[2].root) for l in layers], "final": final_l},
        "queries": queries,
        **({"pow_nonce": pow_nonce} if pow_nonce is not None else {}),
    }
    tm["total"] = time.perf_counter() - t_all
    return proof


def proof_to_json(proof):
    return json.dumps(proof, separators=(",", ":"))
