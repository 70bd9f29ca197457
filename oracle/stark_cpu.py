"""oracle/stark_cpu.py -- CPU backend of the STARK prover built from the oracle primitives
(TEST INFRASTRUCTURE: used by tests/ for bit-exact proof comparison and by bench.py's cpu_baseline
leg).  It plugs into the same orchestration (eigen_zeth_amd/stark/prover.py) as the GPU backend,
so a proof produced here and one produced on the MI355X from the same witness must be identical.
Nothing is imported from the product package: the AIR reaches quotient() as its constraint program blob (data) and is
run by the checker's own interpreter (gl_oracle.c: orc_quotient_program).
PARITY UNPINNED with respect to the external reference prover (see gl_oracle.c)."""
from __future__ import annotations

import numpy as np

from . import oracle as O
from .air_program import Program


class Commit:
    def __init__(self, root, tree, ext=None, coef=None):
        self.root, self.tree, self.ext, self.coef = root, tree, ext, coef


class CpuBackend:
    def __init__(self, rc, mds, root32=O.ROOT32_DEFAULT, shift=O.SHIFT_DEFAULT, hash_mode="gl", bn_tables=None):
        """hash_mode "bn128": 16-ary Poseidon-BN254 trees and transcript permutation (oracle/bn254_hash.c); bn_tables =
        (rc, mds, rp) of the t = 17 instance -- configuration, handed in by the caller like the Goldilocks tables"""
        self.rc, self.mds = np.asarray(rc, dtype=np.uint64), np.asarray(mds, dtype=np.uint64)
        self.root32, self.shift = root32, shift
        self._fixed = {}
        self.hash_mode = hash_mode
        if hash_mode == "bn128":
            O.p254_set(17, bn_tables[2], bn_tables[0], bn_tables[1])

    # ---- Merkle trees of the configured hash mode: (root, tree)
    def _tree(self, mat):
        mat = np.ascontiguousarray(mat)
        if self.hash_mode == "bn128":
            tree = O.merkle16_tree(mat)
            return [O._fr_ints(tree[-1])[0]], tree
        tree = O.merkle_commit(mat, self.rc, self.mds)
        return [int(v) for v in tree[-1]], tree

    def poseidon_bn254_perm17(self, state):
        return O.p254_perm([state], 17)[0]

    def sync(self):
        pass

    def poseidon_perm(self, state):
        return [int(v) for v in O.poseidon_perm(np.array([state], dtype=np.uint64), self.rc, self.mds)[0]]

    def publics_digest(self, pubs):
        """root of the tree a long public-input vector is committed as (restated in oracle/stark_verify.py: publics_digest)"""
        from .stark_verify import publics_digest
        return publics_digest(pubs, self.rc, self.mds, self.hash_mode == "bn128")

    def publics_digest_gl(self, pubs):
        from .stark_verify import publics_digest
        return publics_digest(pubs, self.rc, self.mds, False)

    def poseidon_perm_batch(self, states):
        return O.poseidon_perm(np.asarray(states, dtype=np.uint64), self.rc, self.mds)

    def poseidon_trace(self, inputs):
        return O.poseidon_trace(np.asarray(inputs, dtype=np.uint64), self.rc, self.mds)

    def commit_trace(self, trace, logn, logb, extra_cols=0, group=0):
        W = trace.shape[0]
        e1 = O.lde(trace, logb, self.shift, self.root32)
        root, tree = self._tree(e1.reshape(W << group, -1))       # 2^group rows i, i + M', ... per leaf (BN128 mode)
        ext = np.zeros((W + extra_cols, e1.shape[1]), dtype=np.uint64)
        coef = np.zeros((W + extra_cols, trace.shape[1]), dtype=np.uint64)
        ext[:W], coef[:W] = e1, self._scaled_coef(trace)
        c = Commit(root, tree, ext, coef)
        c.trace, c.W = trace, W
        return c

    def _scaled_coef(self, cols):
        """coefficients of p(shift * X): c_i * shift^i -- the vector the orchestration evaluates at z / shift"""
        return O.coset_scaled_coefficients(cols, self.shift, self.root32)

    def column_view(self, mat, col, rows):
        return np.asarray(mat).reshape(-1, rows)[col:]

    def commit_stage2(self, air, c1, chal, logn, logb, group=0):
        W = c1.W
        parts = []
        for st in air.stage2:
            if st["kind"] == "perm":
                parts.append(O.grand_product(c1.trace[st["a"]], c1.trace[st["b"]], chal))
            else:
                parts.append(O.logup_columns(c1.trace[st["a"]], c1.trace[st["t"]], c1.trace[st["m"]], chal))
        z = np.ascontiguousarray(np.concatenate(parts, axis=0))
        c1.ext[W:] = O.lde(z, logb, self.shift, self.root32)
        c1.coef[W:] = self._scaled_coef(z)
        s2 = np.ascontiguousarray(c1.ext[W:])
        root, tree = self._tree(s2.reshape(s2.shape[0] << group, -1))
        return Commit(root, tree)

    def commit_cols(self, cols, M, W):
        mat = np.ascontiguousarray(np.asarray(cols).reshape(W, M))
        root, tree = self._tree(mat)
        return Commit(root, tree)

    def fixed_ext(self, logn, logb, air=None, pubs=None):
        """LDE of every fixed column, materialised as u64[n_fixed][M]: the two boundary selectors, then the sparse periodic
        columns of the statement (decoded from its program blob by the checker's own reader, tiled to N rows, extended like
        any other column -- the definition, no period tricks)"""
        key = (logn, logb)
        if key not in self._fixed:
            N = 1 << logn
            ind = np.zeros((2, N), dtype=np.uint64)
            ind[0, 0] = 1
            ind[1, N - 1] = 1
            self._fixed[key] = O.lde(ind, logb, self.shift, self.root32)
        sel = self._fixed[key]
        prog = Program(air.program()) if air is not None else None
        if prog is None or not prog.fixed_cols:
            return sel
        N = 1 << logn
        extra = np.zeros((len(prog.fixed_cols), N), dtype=np.uint64)
        for k in range(len(prog.fixed_cols)):
            per = np.array(prog.fixed_period(k, [int(v) for v in pubs]), dtype=np.uint64)
            extra[k] = np.tile(per, N // len(per))
        return np.ascontiguousarray(np.concatenate([sel, O.lde(extra, logb, self.shift, self.root32)], axis=0))

    def quotient(self, air, c1, fixed, pubs, apow, zhinv, logn, logb, wlast):
        logm = logn + logb
        M = 1 << logm
        prog = np.ascontiguousarray(np.asarray(air.program(), dtype=np.uint64))   # the statement, as data
        out = np.empty((3, M), dtype=np.uint64)
        pub = np.array(list(pubs) + [0], dtype=np.uint64)
        ap = np.ascontiguousarray(np.array(apow, dtype=np.uint64).reshape(-1))
        zh = np.array(zhinv, dtype=np.uint64)
        rc = O.lib().orc_quotient_program(O._p(prog), prog.size, O._p(np.ascontiguousarray(c1.ext)), O._p(fixed), M, 1 << logb,
                                          O._p(pub), O._p(ap), O._p(zh), self.shift, O.lib().orc_root(self.root32, logm), wlast,
                                          O._p(out))
        if rc != 0:
            raise ValueError("malformed constraint program")
        return out

    def pow_grind(self, seed4, bits):
        return O.pow_grind(seed4, bits, self.rc, self.mds)

    def coset_coefficients(self, planes, logm, W):
        return O.intt(np.asarray(planes).reshape(W, 1 << logm), self.root32)

    def quotient_pieces(self, q, logn, logb, Q):
        """LDEs of the Q degree-<N pieces of the quotient (slices of the coefficient vector of q(shift X)), piece-major"""
        N, M = 1 << logn, 1 << (logn + logb)
        c = self.coset_coefficients(q, logn + logb, 3)
        pad = np.zeros((3 * Q, M), dtype=np.uint64)
        pad[:, :N] = np.stack([c[p, j * N:(j + 1) * N] for j in range(Q) for p in range(3)])
        return O.ntt(pad, self.root32)

    def eval_ext(self, coef, logn, W, point):
        return O.poly_eval_e3_cols(np.ascontiguousarray(np.asarray(coef).reshape(-1, 1 << logn)[:W]), point)

    def ood_evals(self, c1, Wt, q, Wq, q_logn, logn, logb, zeta, zeta_w):
        """out-of-domain evaluations BY THE DEFINITION: interpolate (the coefficients of p(shift X): trace columns from the trace, quotient
        columns from their committed values on the sub-coset of 2^q_logn points), then evaluate the coefficient form at z / shift.  (The
        product reads the extensions in the barycentric form instead: a different route to the same field elements.)"""
        sinv = pow(self.shift, O.P - 2, O.P)
        zs, zws = [v * sinv % O.P for v in zeta], [v * sinv % O.P for v in zeta_w]
        ev_z, ev_zw = self.eval_ext(c1.coef, logn, Wt, zs), self.eval_ext(c1.coef, logn, Wt, zws)
        logm = logn + logb
        sub = np.ascontiguousarray(np.asarray(q).reshape(Wq, 1 << logm)[:, ::1 << (logm - q_logn)])
        ev_q = self.eval_ext(self.coset_coefficients(sub, q_logn, Wq), q_logn, Wq, zs)
        return ev_z, ev_zw, ev_q

    def deep(self, a, Wa, b, Wb, logm, n_next, z, zw, gamma, ev_z, ev_zw):
        return O.deep_quotient(np.ascontiguousarray(np.asarray(a).reshape(-1, 1 << logm)[:Wa]), np.asarray(b).reshape(Wb, 1 << logm), n_next, z, zw,
                               gamma, ev_z, ev_zw, self.shift, self.root32, fast=True)

    def fri_fold(self, planes, logn, logf, beta, shift):
        return O.fri_fold(np.asarray(planes).reshape(3, 1 << logn), logf, beta, shift, self.root32)

    def download(self, d, shape):
        return np.asarray(d).reshape(shape)

    def gather_rows(self, cols, M, W, idx):
        mat = np.asarray(cols).reshape(-1, M)[:W]
        return np.ascontiguousarray(mat[:, np.asarray(idx, dtype=np.int64)].T)

    def open_paths(self, tree, M, idx):
        if self.hash_mode == "bn128":       # per level the 16 digests of the group on the path (missing children = 0)
            out = []
            for j in idx:
                n, off, pos, lv = M, 0, int(j), []
                while n > 1:
                    g0 = (pos // 16) * 16
                    lv.append([O._fr_ints(tree[off + g0 + c])[0] if g0 + c < n else 0 for c in range(16)])
                    off, n, pos = off + n, (n + 15) // 16, pos // 16
                out.append(lv)
            return out
        depth = int(M).bit_length() - 1
        out = np.zeros((len(idx), depth, 4), dtype=np.uint64)
        for i, j in enumerate(idx):
            out[i] = O.merkle_path(tree, int(j))
        return out

    def msm_g1(self, points, scalars):
        from . import naive_bn254 as B1
        return B1.msm([p if p is not None else (0, 0) for p in points], scalars)

    def groth16_prove(self, key, witness, rand):
        """the checker's Groth16 prover: by the trapdoor of the (seeded, test-only) key -- oracle/groth16_trapdoor.py.  key: an object with
        toxic / u / v / l (u64[n][4] arrays) / n_pub; the same group elements the GPU's transforms and MSMs must give"""
        from . import groth16_trapdoor as GT
        ints = lambda a: [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in a]
        return GT.prove(key.toxic, ints(key.u), ints(key.v), ints(key.l), key.n_pub, witness, rand)

    def qap_quotient(self, a_ev, b_ev, c_ev, logm, coset):
        from . import naive
        return naive.qap_quotient(a_ev, b_ev, c_ev)       # definition level; independent of the coset the GPU path evaluates on
