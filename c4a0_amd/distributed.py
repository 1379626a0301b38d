"""Multi-GPU self-play: one process per GPU, games sharded by position in the request list,
no traffic while playing, ONE variable-length all-gather of the finished samples at the end.

Games never interact (reference rust/src/self_play.rs:55-58: one MctsGame per GameMetadata)
and the move RNG is a pure function of (game_id, n_moves) (mcts.rs:215), so a game's samples
do not depend on which rank plays it.  `torch.distributed` backend "nccl" is RCCL on ROCm
(xGMI between the 8 GPUs of a node); the same code runs over "gloo" on CPU tensors in the tests.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch
import torch.distributed as dist

SAMPLE_BYTES = 64


def shard_indices(n_games: int, rank: int, world_size: int) -> np.ndarray:
    """Request-list positions owned by `rank`: i with i % world_size == rank."""
    return np.arange(rank, n_games, world_size, dtype=np.int64)


def all_gather_records(local: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather a per-rank uint8[n_r, 64] record tensor of varying n_r.

    Counts first (one small all_gather), then one all_gather on buffers padded to the largest
    count: fixed-size collectives map onto RCCL's ring/tree all-gather, and a rank's whole
    shard moves as one message per peer link (payload ~1.1 KB per game)."""
    assert local.dtype == torch.uint8 and local.dim() == 2 and local.shape[1] == SAMPLE_BYTES
    world = dist.get_world_size(group)
    dev = local.device
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    padded = torch.zeros((cap, SAMPLE_BYTES), dtype=torch.uint8, device=dev)
    padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return [b[:c] for b, c in zip(bufs, counts)]


def merge_rank_records(per_rank: Sequence[np.ndarray], n_games: int, world_size: int, counts_per_rank: Sequence[np.ndarray]):
    """Interleave the ranks' packed records back into request-list order.

    per_rank[r] holds the records of games r, r+W, r+2W, ... packed in that order and
    counts_per_rank[r][k] the sample count of game r + k*W.  Returns (records, counts)."""
    from .session import SAMPLE_DTYPE

    counts = np.zeros(n_games, dtype=np.uint32)
    for r in range(world_size):
        idx = shard_indices(n_games, r, world_size)
        counts[idx] = counts_per_rank[r][: idx.size]
    offs = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])
    out = np.zeros(int(offs[-1]), dtype=SAMPLE_DTYPE)
    for r in range(world_size):
        idx = shard_indices(n_games, r, world_size)
        recs = per_rank[r]
        o = 0
        for g in idx:
            c = int(counts[g])
            out[offs[g]:offs[g] + c] = recs[o:o + c]
            o += c
    return out, counts
