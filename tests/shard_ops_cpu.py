"""Local-work adapter of eigen_zeth_amd/stark/sharded.py backed by the CPU checker (TEST INFRASTRUCTURE): lets the
world_size-2 gloo tests run a sharded proof without a GPU.  Tensors are torch CPU int64 views of u64 data."""
import numpy as np
import torch

from oracle import oracle as O


def _np(t):
    return np.ascontiguousarray(t.contiguous().numpy().view(np.uint64)) if isinstance(t, torch.Tensor) else np.ascontiguousarray(np.asarray(t, dtype=np.uint64))


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).view(np.int64))


class CpuShardOps:
    device = "cpu"

    def __init__(self, rc, mds, root32=O.ROOT32_DEFAULT, shift=O.SHIFT_DEFAULT):
        self.rc, self.mds = np.asarray(rc, dtype=np.uint64), np.asarray(mds, dtype=np.uint64)
        self.root32, self.shift = root32, shift
        self._fixed = {}

    def sync(self):
        pass

    def empty(self, shape):
        return torch.zeros(shape, dtype=torch.int64)

    def from_host(self, a):
        return _t(a).clone()

    def to_host(self, t):
        return _np(t)

    def poseidon_perm(self, state):
        return [int(v) for v in O.poseidon_perm(np.array([state], dtype=np.uint64), self.rc, self.mds)[0]]

    def hash_pair(self, l, r):
        return self.poseidon_perm(list(l) + list(r) + [0] * 4)[:4]

    def pow_grind(self, seed4, bits):
        return O.pow_grind(seed4, bits, self.rc, self.mds)

    def lde(self, cols, logn, logb):
        x = _np(cols)
        return _t(O.lde(x, logb, self.shift, self.root32)), _t(O.coset_scaled_coefficients(x, self.shift, self.root32))

    def merkle_commit(self, rows, nloc):
        return _t(O.merkle_commit(_np(rows), self.rc, self.mds).reshape(-1))

    def merkle_commit_any(self, cols, M, W):
        return _t(O.merkle_commit(_np(cols).reshape(W, M), self.rc, self.mds).reshape(-1))

    def tree_root(self, tree, nleaves):
        return [int(v) for v in _np(tree).reshape(-1, 4)[2 * nleaves - 2]]

    def grand_product(self, a, b, N, chal):
        return _t(O.grand_product(_np(a), _np(b), chal))

    def logup_columns(self, a, t, m, N, chal):
        return _t(O.logup_columns(_np(a), _np(t), _np(m), chal))

    def fixed_rows(self, logn, logb, r0, nloc):
        key = (logn, logb)
        if key not in self._fixed:
            N = 1 << logn
            ind = np.zeros((2, N), dtype=np.uint64)
            ind[0, 0] = 1
            ind[1, N - 1] = 1
            self._fixed[key] = O.lde(ind, logb, self.shift, self.root32)
        return _t(self._fixed[key][:, r0:r0 + nloc])

    def quotient_rows(self, program, buf, fx, logm, logb, r0, nloc, pubs, apow, zhinv, wlast):
        prog = np.ascontiguousarray(np.asarray(program, dtype=np.uint64))
        b, f = _np(buf), _np(fx)
        out = np.empty((3, nloc), dtype=np.uint64)
        pub = np.array(list(pubs) + [0], dtype=np.uint64)
        ap = np.ascontiguousarray(np.array(apow, dtype=np.uint64).reshape(-1))
        zh = np.array(zhinv, dtype=np.uint64)
        rc = O.lib().orc_quotient_program_rows(O._p(prog), prog.size, O._p(b), b.shape[1], O._p(f), f.shape[1], 1 << logm, 1 << logb, r0, nloc,
                                               O._p(pub), O._p(ap), O._p(zh), self.shift, O.lib().orc_root(self.root32, logm), wlast,
                                               O._p(out), nloc)
        assert rc == 0
        return _t(out)

    def deep_rows(self, a, Wa, b, Wb, logm, r0, nloc, n_next, z, zw, gamma, ev_z, ev_zw):
        A, B = _np(a), _np(b)
        out = np.empty((3, nloc), dtype=np.uint64)
        ezw = O._arr(ev_zw) if n_next else np.zeros((1, 3), dtype=np.uint64)
        O.lib().orc_deep_quotient_rows(O._p(A), Wa, A.shape[1], O._p(B), Wb, B.shape[1], logm, r0, nloc, n_next, O._p(O._arr(z)),
                                       O._p(O._arr(zw)), O._p(O._arr(gamma)), O._p(O._arr(ev_z)), O._p(ezw), self.shift, self.root32,
                                       O._p(out), nloc)
        return _t(out)

    def intt(self, cols, logn):
        return _t(O.intt(_np(cols).reshape(-1, 1 << logn), self.root32))

    def split_quotient(self, coef, logn, logb, Q):
        N, M = 1 << logn, 1 << (logn + logb)
        c = _np(coef).reshape(3, M)
        pieces = np.ascontiguousarray(np.stack([c[p, j * N:(j + 1) * N] for j in range(Q) for p in range(3)]))
        pad = np.zeros((3 * Q, M), dtype=np.uint64)
        pad[:, :N] = pieces
        return _t(O.ntt(pad, self.root32)), _t(pieces)

    def eval_ext(self, coef, logn, W, point):
        return O.poly_eval_e3_cols(_np(coef).reshape(-1, 1 << logn)[:W], point)

    def fri_fold(self, planes, logn, logf, beta, shift):
        return _t(O.fri_fold(_np(planes).reshape(3, 1 << logn), logf, beta, shift, self.root32))

    def gather_rows(self, cols, M, W, idx):
        mat = _np(cols).reshape(-1, M)[:W]
        return np.ascontiguousarray(mat[:, np.asarray(idx, dtype=np.int64)].T)

    def open_paths(self, tree, M, idx):
        tr = _np(tree).reshape(-1, 4)
        depth = int(M).bit_length() - 1
        out = np.zeros((len(idx), max(depth, 1), 4), dtype=np.uint64)
        for i, j in enumerate(idx):
            if depth:
                out[i, :depth] = O.merkle_path(tr, int(j))
        return out[:, :depth] if depth else out[:, :0]
