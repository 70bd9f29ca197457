"""One chunk STARK spread over the GPUs of a node, past the commitment stage.

The orchestration (stark/prover.py) is unchanged: it talks to a backend.  ShardedBackend is that backend for G ranks
(one process per GPU, torch.distributed: RCCL on GPUs, gloo in the CPU tests); every rank runs the same prove() call,
draws the same Fiat-Shamir challenges and ends with the same proof, byte for byte equal to the single-GPU proof.

    stage                       sharding                                   exchange
    trace LDE                   columns  [W/G][N] -> [W/G][M]              none
    trace commitment            rows     [W][M/G]: local subtree           ONE all-to-all (columns -> rows), all-gather of G sub-roots
    stage-2 columns             replicated (a handful of columns)          all-gather of the witness columns they read
    constraint quotient         rows, with a blow-up halo (b rows)         all-gather of G x b halo rows
    quotient commitment         rows: local subtree                        all-gather of G sub-roots
    out-of-domain evaluations   columns (coefficients never moved)         all-gather of the evaluations
    DEEP quotient               rows                                       all-gather of the result (3 columns) -- "gather before FRI"
    FRI, proof of work          replicated                                 none
    query openings              owner of each row answers                  all-gather of rows and sub-tree paths

Local work goes through a small `ops` adapter: HipShardOps (this file: torch CUDA tensors + the C-ABI) on GPUs; the CPU
tests plug the checker's restatement in through the same interface (tests/shard_ops_cpu.py).  Serves GenChunkProof
(src/prover/provider.rs:358-390) for traces too large, or too urgent, for one GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .. import multigpu as MG
from . import air as air_mod
from . import field as F

P = F.P


def _u(t):
    """torch int64 tensor -> list of python ints in [0, 2^64)"""
    return [int(v) & 0xFFFFFFFFFFFFFFFF for v in t.reshape(-1).tolist()]


def _s(vals, device):
    return torch.tensor([int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v) for v in vals], dtype=torch.int64, device=device)


class ShardMat:
    """rows [row0, row0 + nloc) of a [W][M] column-major matrix: local tensor [W][stride >= nloc]"""

    def __init__(self, t, M, row0, nloc, col0=0, ncols=None):
        self.t, self.M, self.row0, self.nloc, self.col0 = t, M, row0, nloc, col0
        self.ncols = t.shape[0] - col0 if ncols is None else ncols


class ShardTree:
    """local subtree over nloc leaves + the G sub-roots (every rank holds the top of the tree)"""

    def __init__(self, local_tree, nloc, subroots):
        self.local, self.nloc, self.subroots = local_tree, nloc, subroots


class ShardCoef:
    """coefficient columns: this rank's trace columns [col0, col0 + wl) and (replicated) the stage-2 columns"""

    def __init__(self, local, col0, W, s2=None):
        self.local, self.col0, self.W, self.s2 = local, col0, W, s2


class Commit:
    def __init__(self, root, tree, ext=None, coef=None):
        self.root, self.tree, self.ext, self.coef = root, tree, ext, coef


class ShardedBackend:
    def __init__(self, ops, group=None):
        self.ops, self.group = ops, group
        self.G = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.shift, self.root32 = ops.shift, ops.root32
        self.dev = ops.device

    # ---- collectives
    def _all_gather(self, t):
        out = [torch.empty_like(t) for _ in range(self.G)]
        MG.all_gather(out, t.contiguous(), group=self.group)
        return out

    def _cols_to_rows(self, local_cols):
        """[wl][M] (my columns, all rows) -> [G*wl][M/G] (all columns, my rows): one all-to-all"""
        from .. import multigpu
        rows, _ = multigpu.exchange_columns_to_rows(local_cols, self.group)
        return rows

    def _top(self, subroots):
        lvl, levels = [list(r) for r in subroots], []
        while len(lvl) > 1:
            levels.append(lvl)
            lvl = [self.ops.hash_pair(lvl[2 * i], lvl[2 * i + 1]) for i in range(len(lvl) // 2)]
        return lvl[0], levels     # root, [level of G nodes, level of G/2 nodes, ...]

    def _commit_rows(self, rows, nloc):
        """local subtree of rows [W][nloc] + all-gather of the sub-roots -> (root, ShardTree)"""
        tree = self.ops.merkle_commit(rows, nloc)
        sub = self.ops.tree_root(tree, nloc)
        subroots = [_u(t) for t in self._all_gather(_s(sub, self.dev))]
        root, _ = self._top(subroots)
        return root, ShardTree(tree, nloc, subroots)

    # ---- backend interface of stark/prover.py
    def sync(self):
        self.ops.sync()

    def poseidon_perm(self, state):
        return self.ops.poseidon_perm(state)

    def pow_grind(self, seed4, bits):
        return self.ops.pow_grind(seed4, bits)

    def commit_trace(self, trace, logn, logb, extra_cols=0, group=0):
        assert group == 0, "Goldilocks mode only"
        """trace: the witness [W][N]; a rank reads only its W/G columns of it (plus, in stage 2, the few columns the
        permutation / lookup arguments name)"""
        N, M = 1 << logn, 1 << (logn + logb)
        G, nloc = self.G, M // self.G
        full = np.asarray(trace)
        W = full.shape[0]
        assert W % G == 0 and nloc % (1 << logb) == 0 and nloc >= (1 << logb), "columns and rows must split evenly over the ranks"
        wl = W // G
        mine = full[self.rank * wl:(self.rank + 1) * wl]      # this rank's column shard; the other columns are never touched here
        local = self.ops.from_host(mine)
        ext_cols, coef = self.ops.lde(local, logn, logb)                    # [wl][M], [wl][N]: no communication
        rows = self._cols_to_rows(ext_cols)                                 # [W][nloc]
        del ext_cols
        ext = self.ops.empty((W + extra_cols, nloc))
        ext[:W] = rows
        root, tree = self._commit_rows(ext[:W], nloc)
        c = Commit(root, tree, ShardMat(ext, M, self.rank * nloc, nloc, 0, W), ShardCoef(coef, self.rank * wl, W))
        c.W, c.logn, c.witness = W, logn, full
        c.local_witness, c.wl = local, wl
        return c

    def column_view(self, mat, col, rows):
        assert isinstance(mat, ShardMat)
        return ShardMat(mat.t, mat.M, mat.row0, mat.nloc, col, mat.t.shape[0] - col)

    def _witness_column(self, c1, idx, N):
        """column idx of the witness on every rank (all-gather from its owner)"""
        if c1.witness is not None:
            return self.ops.from_host(c1.witness[idx:idx + 1])[0]
        owner, j = divmod(idx, c1.wl)
        t = c1.local_witness[j] if owner == self.rank else self.ops.empty((N,))
        MG.broadcast(t, owner, group=self.group)
        return t

    def commit_stage2(self, air, c1, chal, logn, logb, group=0):
        assert group == 0, "Goldilocks mode only"
        N, M, W, W2 = 1 << logn, 1 << (logn + logb), c1.W, air.width2
        nloc = M // self.G
        parts = []
        for st in air.stage2:     # a handful of columns: computed on every rank (a sequential scan, not worth an exchange)
            col = lambda k: self._witness_column(c1, st[k], N)
            if st["kind"] == "perm":
                parts.append(self.ops.grand_product(col("a"), col("b"), N, chal))
            else:
                parts.append(self.ops.logup_columns(col("a"), col("t"), col("m"), N, chal))
        s2 = torch.cat(parts, dim=0)
        ext2, coef2 = self.ops.lde(s2, logn, logb)                           # replicated: W2 << W
        r0 = self.rank * nloc
        c1.ext.t[W:W + W2] = ext2[:, r0:r0 + nloc]
        c1.ext.ncols = W + W2
        c1.coef.s2 = coef2
        root, tree = self._commit_rows(c1.ext.t[W:W + W2], nloc)
        return Commit(root, tree)

    def commit_cols(self, cols, M, W):
        if isinstance(cols, ShardMat):
            root, tree = self._commit_rows(cols.t[cols.col0:cols.col0 + W, :cols.nloc], cols.nloc)
            return Commit(root, tree)
        tree = self.ops.merkle_commit_any(cols, M, W)
        return Commit(self.ops.tree_root(tree, M), tree)

    def fixed_ext(self, logn, logb):
        return ("fixed", logn, logb)

    def quotient(self, air, c1, fixed, pubs, apow, zhinv, logn, logb, wlast):
        logm, b = logn + logb, 1 << logb
        M, nloc, r0 = 1 << logm, c1.ext.nloc, c1.ext.row0
        Wt = c1.ext.ncols
        # halo: the first b rows of every shard, all-gathered (G * Wt * b elements); mine are the next rank's
        heads = self._all_gather(c1.ext.t[:Wt, :b].contiguous())
        buf = self.ops.empty((Wt, nloc + b))
        buf[:, :nloc] = c1.ext.t[:Wt, :nloc]
        buf[:, nloc:] = heads[(self.rank + 1) % self.G]
        fx = self.ops.fixed_rows(logn, logb, r0, nloc)                       # [2][nloc] of the replicated selector LDEs
        q = self.ops.quotient_rows(air.program(), buf, fx, logm, logb, r0, nloc, list(pubs), apow, zhinv, wlast)
        return ShardMat(q, M, r0, nloc, 0, 3)

    def coset_coefficients(self, planes, logm, W):
        if isinstance(planes, ShardMat):
            full = torch.cat(self._all_gather(planes.t[planes.col0:planes.col0 + W, :planes.nloc].contiguous()), dim=1)
            return self.ops.intt(full, logm)
        return self.ops.intt(planes, logm)

    def split_quotient(self, coef, logn, logb, Q):
        """the quotient's coefficients are replicated after coset_coefficients: every rank forms the Q pieces (3Q short
        columns) and keeps its rows of their LDEs"""
        ext, pieces = self.ops.split_quotient(coef, logn, logb, Q)
        M = 1 << (logn + logb)
        nloc = M // self.G
        r0 = self.rank * nloc
        return ShardMat(ext[:, r0:r0 + nloc].contiguous(), M, r0, nloc, 0, 3 * Q), pieces

    def quotient_pieces(self, q, logn, logb, Q):
        """stark/prover.py's hook: the LDEs of the quotient's Q pieces (my rows); their coefficient slices are kept for ood_evals"""
        ext, self._qcoef = self.split_quotient(self.coset_coefficients(q, logn + logb, 3), logn, logb, Q)
        return ext

    def ood_evals(self, c1, Wt, d_q, Wq, q_logn, logn, logb, zeta, zeta_w):
        """this torch.distributed orchestration stays on the coefficient route (every rank evaluates the coefficient columns it owns,
        the evaluations are all-gathered); the C++ sharded prover (csrc/prove.hip) reads values instead (zp_ood_eval)"""
        sinv = F.inv(self.shift)
        zs, zws = F.e3_scale(zeta, sinv), F.e3_scale(zeta_w, sinv)
        ev_z, ev_zw = self.eval_ext(c1.coef, logn, Wt, zs), self.eval_ext(c1.coef, logn, Wt, zws)
        qcoef = self._qcoef if q_logn == logn and Wq > 3 else self.coset_coefficients(d_q, logn + logb, 3)
        return ev_z, ev_zw, self.eval_ext(qcoef, q_logn, Wq, zs)

    def eval_ext(self, coef, logn, W, point):
        if not isinstance(coef, ShardCoef):
            return self.ops.eval_ext(coef, logn, W, point)
        mine = self.ops.eval_ext(coef.local, logn, coef.local.shape[0], point)         # [wl][3]
        parts = self._all_gather(_s(np.asarray(mine, dtype=np.uint64).reshape(-1), self.dev))
        out = [v for t in parts for v in _u(t)]
        rows = [out[3 * i:3 * i + 3] for i in range(coef.W)]
        if W > coef.W:
            rows += [list(map(int, r)) for r in self.ops.eval_ext(coef.s2, logn, W - coef.W, point)]
        return rows

    def deep(self, a, Wa, b, Wb, logm, n_next, z, zw, gamma, ev_z, ev_zw):
        assert isinstance(a, ShardMat) and isinstance(b, ShardMat)
        f = self.ops.deep_rows(a.t[:Wa, :a.nloc], Wa, b.t[b.col0:b.col0 + Wb, :b.nloc], Wb, logm, a.row0, a.nloc, n_next, z, zw, gamma,
                               ev_z, ev_zw)
        return torch.cat(self._all_gather(f.contiguous()), dim=1).contiguous()         # [3][M] on every rank: FRI is replicated

    def fri_fold(self, planes, logn, logf, beta, shift):
        return self.ops.fri_fold(planes, logn, logf, beta, shift)

    def download(self, d, shape):
        return self.ops.to_host(d).reshape(shape)

    def gather_rows(self, cols, M, W, idx):
        if not isinstance(cols, ShardMat):
            return self.ops.gather_rows(cols, M, W, idx)
        mine = torch.zeros((len(idx), W), dtype=torch.int64, device=self.dev)
        for i, j in enumerate(idx):
            if cols.row0 <= j < cols.row0 + cols.nloc:
                mine[i] = cols.t[cols.col0:cols.col0 + W, j - cols.row0]
        MG.all_reduce(mine, group=self.group)      # every row has exactly one owner; the others contribute zeros
        return self.ops.to_host(mine).reshape(len(idx), W)

    def open_paths(self, tree, M, idx):
        if not isinstance(tree, ShardTree):
            return self.ops.open_paths(tree, M, idx)
        dl = tree.nloc.bit_length() - 1
        depth = int(M).bit_length() - 1
        mine = torch.zeros((len(idx), max(dl, 1), 4), dtype=torch.int64, device=self.dev)
        own = [i for i, j in enumerate(idx) if self.rank * tree.nloc <= j < (self.rank + 1) * tree.nloc]
        if own and dl > 0:
            paths = self.ops.open_paths(tree.local, tree.nloc, [idx[i] - self.rank * tree.nloc for i in own])
            mine[own] = self.ops.from_host(np.asarray(paths, dtype=np.uint64).reshape(len(own), dl, 4))
        MG.all_reduce(mine, group=self.group)
        low = self.ops.to_host(mine).reshape(len(idx), max(dl, 1), 4)
        _, levels = self._top(tree.subroots)
        out = np.zeros((len(idx), depth, 4), dtype=np.uint64)
        for i, j in enumerate(idx):
            out[i, :dl] = low[i, :dl]
            node = j >> dl
            for l, lvl in enumerate(levels):
                out[i, dl + l] = np.array(lvl[node ^ 1], dtype=np.uint64)
                node >>= 1
        return out


class HipShardOps:
    """local work of a shard on one MI355X: torch CUDA tensors (int64 storage of u64 values) through the C-ABI.
    The Prover must run on torch's current stream (Prover(dev, stream=torch.cuda.current_stream().cuda_stream))."""

    def __init__(self, prover, device):
        from .. import native
        self.p, self.device = prover, device
        self.root32 = int(prover.get_constants(native.ZP_CONST_ROOT32, 1)[0])
        self.shift = int(prover.get_constants(native.ZP_CONST_COSET_SHIFT, 1)[0])
        self._fixed = {}
        self._st = torch.zeros((12,), dtype=torch.int64, device=device)

    def sync(self):
        self.p.sync()

    def empty(self, shape):
        return torch.empty(shape, dtype=torch.int64, device=self.device)

    def from_host(self, a):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.uint64)).view(np.int64)).to(self.device)

    def to_host(self, t):
        if isinstance(t, torch.Tensor):
            return t.contiguous().cpu().numpy().view(np.uint64)
        return np.asarray(t, dtype=np.uint64)

    def poseidon_perm(self, state):
        self._st.copy_(_s(state, "cpu"))
        self.p.poseidon_perm(self._st, 1)
        return _u(self._st)

    def hash_pair(self, l, r):
        return self.poseidon_perm(list(l) + list(r) + [0] * 4)[:4]

    def pow_grind(self, seed4, bits):
        return self.p.pow_grind(seed4, bits)

    def lde(self, cols, logn, logb):
        w = cols.shape[0]
        ext, coef = self.empty((w, 1 << (logn + logb))), self.empty((w, 1 << logn))
        self.p.lde(cols.contiguous(), ext, logn, logb, w, self.shift, d_coef=coef)
        return ext, coef

    def merkle_commit(self, rows, nloc):
        W = rows.shape[0]
        tree = self.empty(((2 * nloc - 1) * 4,))
        self.p.merkle_commit(rows.contiguous(), nloc, W, tree)
        return tree

    def merkle_commit_any(self, cols, M, W):
        tree = self.empty(((2 * M - 1) * 4,))
        self.p.merkle_commit(cols, M, W, tree)
        return tree

    def tree_root(self, tree, nleaves):
        return _u(tree[(2 * nleaves - 2) * 4:(2 * nleaves - 1) * 4])

    def grand_product(self, a, b, N, chal):
        out = self.empty((3, N))
        self.p.grand_product(a.contiguous(), b.contiguous(), N, chal, out)
        return out

    def logup_columns(self, a, t, m, N, chal):
        out = self.empty((9, N))
        self.p.logup_columns(a.contiguous(), t.contiguous(), m.contiguous(), N, chal, out)
        return out

    def fixed_rows(self, logn, logb, r0, nloc):
        key = (logn, logb)
        if key not in self._fixed:
            N = 1 << logn
            ind = np.zeros((2, N), dtype=np.uint64)
            ind[0, 0] = 1
            ind[1, N - 1] = 1
            self._fixed[key] = self.lde(self.from_host(ind), logn, logb)[0]
        return self._fixed[key][:, r0:r0 + nloc].contiguous()

    def quotient_rows(self, program, buf, fx, logm, logb, r0, nloc, pubs, apow, zhinv, wlast):
        out = self.empty((3, nloc))
        self.p.eval_quotient_rows(program, buf, buf.shape[1], fx, fx.shape[1], logm, logb, r0, nloc, pubs, apow, zhinv, self.shift, wlast,
                                  out, nloc)
        return out

    def deep_rows(self, a, Wa, b, Wb, logm, r0, nloc, n_next, z, zw, gamma, ev_z, ev_zw):
        a, b = a.contiguous(), b.contiguous()
        out = self.empty((3, nloc))
        self.p.deep_quotient_rows(a, Wa, a.shape[1], b, Wb, b.shape[1], logm, r0, nloc, n_next, z, zw, gamma, ev_z, ev_zw, self.shift, out, nloc)
        return out

    def intt(self, cols, logn):
        cols = cols.contiguous()
        out = torch.empty_like(cols)
        self.p.intt(cols, out, logn, cols.shape[0])
        return out

    def split_quotient(self, coef, logn, logb, Q):
        N, M = 1 << logn, 1 << (logn + logb)
        c = coef.reshape(3, M)
        pieces = torch.stack([c[p, j * N:(j + 1) * N] for j in range(Q) for p in range(3)]).contiguous()
        pad = torch.zeros((3 * Q, M), dtype=torch.int64, device=self.device)
        pad[:, :N] = pieces
        ext = torch.empty_like(pad)
        self.p.ntt(pad, ext, logn + logb, 3 * Q)
        return ext, pieces

    def eval_ext(self, coef, logn, W, point):
        return self.p.poly_eval_ext(coef, logn, W, point)

    def fri_fold(self, planes, logn, logf, beta, shift):
        out = self.empty((3, 1 << (logn - logf)))
        self.p.fri_fold(planes, out, logn, logf, beta, shift)
        return out

    def gather_rows(self, cols, M, W, idx):
        return self.p.gather_rows(cols, M, W, idx)

    def open_paths(self, tree, M, idx):
        return self.p.merkle_open_batch(tree, M, idx)
