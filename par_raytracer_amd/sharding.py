"""Interleaved scan-line-block sharding of a frame over ranks and its re-assembly.

The reference gives MPI rank r the contiguous pixel range [r*cpp, (r+1)*cpp) and MPI_Gathers float RGBA to
rank 0 (main.cpp:311-347); contiguous ranges balance badly (NOTES.txt:25), so here rank r of n renders the
blocks of `block_rows` rows whose block index b satisfies b % n == r, packed densely in ascending row order
(the same walk as prt_shard_rows / prt_render_shard_device in include/prt.h).  The gather needs equal
shard sizes, so shards are padded to the largest one; `row_index` maps every (rank, local row) slot of the
gathered [n * max_rows] stack to its image row, padding slots to a scratch row `height`.
"""
from __future__ import annotations

from typing import List

import numpy as np


def shard_row_list(height: int, block_rows: int, rank: int, nranks: int) -> np.ndarray:
    """Image rows rendered by `rank`, in the order they are packed in its shard."""
    rows: List[int] = []
    b = rank
    while b * block_rows < height:
        y0 = b * block_rows
        rows.extend(range(y0, min(y0 + block_rows, height)))
        b += nranks
    return np.asarray(rows, dtype=np.int64)


def shard_rows(height: int, block_rows: int, rank: int, nranks: int) -> int:
    return int(len(shard_row_list(height, block_rows, rank, nranks)))


def max_shard_rows(height: int, block_rows: int, nranks: int) -> int:
    return max(shard_rows(height, block_rows, r, nranks) for r in range(nranks))


def row_index(height: int, block_rows: int, nranks: int) -> np.ndarray:
    """[nranks * max_rows] destination row of every gathered slot (padding -> `height`)."""
    m = max_shard_rows(height, block_rows, nranks)
    idx = np.full((nranks, m), height, dtype=np.int64)
    for r in range(nranks):
        rows = shard_row_list(height, block_rows, r, nranks)
        idx[r, :len(rows)] = rows
    return idx.reshape(-1)


def assemble(gathered, row_idx, height: int):
    """gathered: torch tensor [nranks * max_rows, width, C]; returns [height, width, C] on the same device."""
    import torch
    out = torch.empty((height + 1,) + tuple(gathered.shape[1:]), dtype=gathered.dtype, device=gathered.device)
    out.index_copy_(0, row_idx, gathered)
    return out[:height]
