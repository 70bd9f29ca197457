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


def pack_for_exchange(local_cols, G):
    """[Wl][M] -> [G][Wl][M/G] contiguous (block h = the rows that go to rank h)"""
    Wl, M = local_cols.shape
    return local_cols.view(Wl, G, M // G).permute(1, 0, 2).contiguous()


def exchange_columns_to_rows(local_cols, group=None):
    """all-to-all: returns [G*Wl][M/G] (all columns, local rows), and the bytes this rank sent"""
    G = dist.get_world_size(group)
    Wl, M = local_cols.shape
    send = pack_for_exchange(local_cols, G)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
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
    dist.all_gather(allr, t, group=group)
    subroots = [[int(v) & 0xFFFFFFFFFFFFFFFF for v in r.tolist()] for r in allr]
    return tree_top(subroots, hash_pair), {"sent_bytes": sent, "rows_shape": tuple(rows.shape)}
