"""Multi-GPU commit: column shards -> (one all-to-all over xGMI) -> row shards -> local Merkle subtrees
-> all-gather of the sub-roots -> top of the tree.  SURVEY.md 8e.

NTT / LDE act on columns independently, so ranks own W/G columns and transform them with no
communication.  Leaf hashing needs whole rows, so there is exactly one exchange: rank g sends rank h
the rows [h*M/G, (h+1)*M/G) of its columns.  With the send buffer packed as [G][W/G][M/G] the
received buffer is [G*W/G][M/G] = the column-major matrix of ALL columns restricted to the local
rows, i.e. exactly what zp_merkle_commit takes -- no unpack pass.  xGMI is point-to-point, so the
all-to-all keeps all 7 links of every GPU busy at once (RCCL all_to_all_single = grouped send/recv).

One process per GPU (torch.distributed, backend "nccl" = RCCL); tensors are int64 views of u64 data.
The same code runs on CPU tensors over gloo with a CPU backend for the world_size-2 tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


_LAYOUT = {"prover": None}


# ---- collectives.  On GPUs the backend is "nccl" (= RCCL over xGMI) and tensors go in as they are.  The same code also runs
# with the gloo backend: on CPU tensors (the world-2 CPU tests) and -- for rehearsing several ranks on ONE GPU, which RCCL
# refuses -- on device tensors staged through the host (data path only: HIP kernels, layouts and indexing are the real ones).
def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def all_to_all_single(recv, send, group=None):
    if _staged(send, group):
        r = torch.empty(send.shape, dtype=send.dtype)
        dist.all_to_all_single(r, send.cpu(), group=group)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group)


def all_gather(outs, t, group=None):
    if _staged(t, group):
        tmp = [torch.empty(t.shape, dtype=t.dtype) for _ in outs]
        dist.all_gather(tmp, t.cpu(), group=group)
        for o, v in zip(outs, tmp):
            o.copy_(v)
    else:
        dist.all_gather(outs, t, group=group)


def all_reduce(t, group=None, op=None):
    op = dist.ReduceOp.SUM if op is None else op
    if _staged(t, group):
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)


def broadcast(t, src, group=None):
    if _staged(t, group):
        c = t.cpu()
        dist.broadcast(c, src, group=group)
        t.copy_(c)
    else:
        dist.broadcast(t, src, group=group)


def use_device_layout(prover):
    """route the packing / transposing copies of CUDA tensors through the library's HIP kernels (zp_pack_blocks,
    zp_transpose) on `prover`'s stream instead of generic tensor copies; None switches back (CPU tensors always use torch)"""
    _LAYOUT["prover"] = prover


def _hip(t):
    return _LAYOUT["prover"] if (t.is_cuda and _LAYOUT["prover"] is not None) else None


def pack_for_exchange(local_cols, G):
    """[Wl][M] -> [G][Wl][M/G] contiguous (block h = the rows that go to rank h)"""
    Wl, M = local_cols.shape
    p = _hip(local_cols)
    if p is not None and (M // G) % 2 == 0:
        out = torch.empty((G, Wl, M // G), dtype=local_cols.dtype, device=local_cols.device)
        p.pack_blocks(local_cols.contiguous(), out, Wl, M, G)
        return out
    return local_cols.view(Wl, G, M // G).permute(1, 0, 2).contiguous()


def transpose2d(mat):
    """[R][C] -> [C][R] contiguous"""
    p = _hip(mat)
    if p is not None:
        R, C = mat.shape
        out = torch.empty((C, R), dtype=mat.dtype, device=mat.device)
        p.transpose(mat.contiguous(), out, R, C)
        return out
    return mat.t().contiguous()


def exchange_columns_to_rows(local_cols, group=None):
    """all-to-all: returns [G*Wl][M/G] (all columns, local rows), and the bytes this rank sent"""
    G = dist.get_world_size(group)
    Wl, M = local_cols.shape
    send = pack_for_exchange(local_cols, G)
    recv = torch.empty_like(send)
    all_to_all_single(recv, send, group=group)
    sent_bytes = send.numel() * 8 * (G - 1) // G
    return recv.view(G * Wl, M // G), sent_bytes


def tree_top(subroots, hash_pair):
    """subroots: list of G 4-element roots (rank order) -> global root; hash_pair(l, r) -> 4 elements"""
    lvl = [list(r) for r in subroots]
    while len(lvl) > 1:
        lvl = [hash_pair(lvl[2 * i], lvl[2 * i + 1]) for i in range(len(lvl) // 2)]
    return lvl[0]


def distributed_commit(local_ext, commit_rows_fn, hash_pair, group=None):
    """local_ext: [Wl][M] tensor (this rank's LDE columns).  commit_rows_fn(matrix [W][M/G]) -> 4-element
    sub-root (list of ints) of the local rows.  Returns (global_root, stats)."""
    G = dist.get_world_size(group)
    rows, sent = exchange_columns_to_rows(local_ext, group)
    sub = commit_rows_fn(rows)
    t = torch.tensor([int(v) - (1 << 64) if int(v) >= (1 << 63) else int(v) for v in sub], dtype=torch.int64,
                     device=local_ext.device)
    allr = [torch.empty_like(t) for _ in range(G)]
    all_gather(allr, t, group=group)
    subroots = [[int(v) & 0xFFFFFFFFFFFFFFFF for v in r.tolist()] for r in allr]
    return tree_top(subroots, hash_pair), {"sent_bytes": sent, "rows_shape": tuple(rows.shape)}


# ---------------------------------------------------------------------------------------------------------
# One column split over GPUs ("single huge column"): four-step NTT.  N = N1*N2, input index
# i = i1*N2 + i2, output index k = k1 + N1*k2:
#     X[k1 + N1 k2] = sum_i2 w_N2^(i2 k2) * w_N^(i2 k1) * ( sum_i1 x[i1 N2 + i2] w_N1^(i1 k1) )
# Rank g holds the contiguous block i in [g N/G, (g+1) N/G) = rows i1 of the N1 x N2 matrix.  Steps: transpose
# (all-to-all) -> N1-point transforms on the local N2/G rows -> twiddle w_N^(i2 k1) -> transpose (all-to-all) ->
# N2-point transforms on the local N1/G rows -> (optionally) a third transpose back to natural order.
# Every transpose is one all_to_all_single with the same message shape as the column->row exchange above.

def distributed_transpose(local, group=None):
    """local [R/G][C] (this rank's rows of an R x C matrix) -> [C/G][R] (this rank's rows of the transpose)"""
    G = dist.get_world_size(group) if dist.is_initialized() else 1
    Rl, C = local.shape
    if G == 1:
        return transpose2d(local)
    send = pack_for_exchange(local, G)                                      # block h = my rows, rank h's columns
    recv = torch.empty_like(send)
    all_to_all_single(recv, send, group=group)                         # recv[h] = rank h's rows, my columns
    return transpose2d(recv.view(G * Rl, C // G))                           # [my column][global row h*Rl + r]


def four_step_ntt(local, logn, ntt_rows, twiddle_rows, group=None, inverse=False, natural_output=True):
    """local: int64/uint64 tensor of N/G elements, this rank's contiguous block of the column (N = 2^logn).
    ntt_rows(mat [W][n], inverse) -> transformed rows (natural order in/out, as zp_ntt / zp_intt);
    twiddle_rows(mat [W][n], row0, logn_total, inverse) -> mat[r][k] * w_N^((row0+r) k)   (zp_twiddle_rows).
    Returns this rank's contiguous block of the transform if natural_output, else the N1/G rows k1 of
    Y[k1][k2] = X[k1 + N1 k2]."""
    G = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    l1 = logn // 2
    l2 = logn - l1
    N1, N2 = 1 << l1, 1 << l2
    assert N1 % G == 0 and N2 % G == 0 and local.numel() == (N1 // G) * N2
    a = distributed_transpose(local.view(N1 // G, N2), group)                # [N2/G][N1]: row i2, entries over i1
    b = ntt_rows(a, inverse)                                                 # over i1 -> k1
    b = twiddle_rows(b, rank * (N2 // G), logn, inverse)                     # * w_N^(i2 k1)
    c = distributed_transpose(b, group)                                      # [N1/G][N2]: row k1, entries over i2
    d = ntt_rows(c, inverse)                                                 # over i2 -> k2 : Y[k1][k2]
    if not natural_output:
        return d
    return distributed_transpose(d, group).reshape(-1)                       # [N2/G][N1]: k = k1 + N1 k2, my k2 rows


def hip_row_ops(prover):
    """(ntt_rows, twiddle_rows) for four_step_ntt on torch CUDA tensors through the C-ABI (zp_ntt / zp_intt /
    zp_twiddle_rows).  The prover's ctx must run on torch's current stream (Prover(dev, stream=...)).  Also routes the
    packing / transposing copies of CUDA tensors through the library's layout kernels (use_device_layout)."""
    use_device_layout(prover)

    def ntt_rows(mat, inverse):
        W, n = mat.shape
        out = torch.empty_like(mat)
        (prover.intt if inverse else prover.ntt)(mat, out, n.bit_length() - 1, W)
        return out

    def twiddle_rows(mat, row0, logn_total, inverse):
        W, n = mat.shape
        prover.twiddle_rows(mat, n.bit_length() - 1, W, row0, logn_total, inverse)
        return mat

    return ntt_rows, twiddle_rows


# ---------------------------------------------------------------------------------------------------------
# MSM over several GPUs: independent point ranges per rank, one affine partial sum each, combined on
# every rank -- "replicas + trivial reduce", no data-path collective beyond the all-gather of G x 64 bytes.

def distributed_msm(local_msm, add_points, group=None):
    """local_msm() -> this rank's partial sum as (x, y) ints or None (infinity), e.g. Prover.msm_bn254 over the rank's
    range of points and scalars; add_points(p, q) -> p + q on the curve (None = infinity).  Returns the total."""
    part = local_msm()
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return part
    G = dist.get_world_size(group)
    words = [0] * 17                      # flag + x, y as 8 x 32-bit words each (int64 tensor elements)
    if part is not None:
        words[0] = 1
        for k in range(8):
            words[1 + k] = (part[0] >> (32 * k)) & 0xFFFFFFFF
            words[9 + k] = (part[1] >> (32 * k)) & 0xFFFFFFFF
    t = torch.tensor(words, dtype=torch.int64)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    allr = [torch.empty_like(t) for _ in range(G)]
    all_gather(allr, t, group=group)
    total = None
    for r in allr:
        w = r.tolist()
        if w[0]:
            p = (sum(w[1 + k] << (32 * k) for k in range(8)), sum(w[9 + k] << (32 * k) for k in range(8)))
            total = add_points(total, p)
    return total
