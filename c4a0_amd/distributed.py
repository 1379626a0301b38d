"""Multi-GPU self-play: one process per GPU, games sharded by position in the request list,
no traffic while playing, ONE variable-length all-gather of the finished samples at the end.

Games never interact (reference rust/src/self_play.rs:55-58: one MctsGame per GameMetadata)
and the move RNG is a pure function of (game_id, n_moves) (mcts.rs:215), so a game's samples
do not depend on which rank plays it.  `torch.distributed` backend "nccl" is RCCL on ROCm
(xGMI between the 8 GPUs of a node); the same code runs over "gloo" (records staged through the
host) in the tests.

    play_games_sharded(reqs, ...)   the product entry point: shard -> play -> pack -> gather -> merge
    shard_indices / gather_shards / merge_shards   its pieces
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

SAMPLE_BYTES = 64


def shard_indices(n_games: int, rank: int, world_size: int) -> np.ndarray:
    """Request-list positions owned by `rank`: i with i % world_size == rank."""
    return np.arange(rank, n_games, world_size, dtype=np.int64)


def _collective_device(local: torch.Tensor, group=None) -> torch.device:
    """gloo moves host memory: stage through the CPU; RCCL ("nccl") takes the device tensor."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else local.device


def all_gather_records(local: torch.Tensor, group=None) -> List[torch.Tensor]:
    """All-gather a per-rank uint8[n_r, 64] record tensor of varying n_r.

    Counts first (one small all_gather), then one all_gather on buffers padded to the largest
    count: fixed-size collectives map onto RCCL's ring/tree all-gather, and a rank's whole
    shard moves as one message per peer link (payload ~1.1 KB per game)."""
    assert local.dtype == torch.uint8 and local.dim() == 2 and local.shape[1] == SAMPLE_BYTES
    world = dist.get_world_size(group)
    dev = _collective_device(local, group)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n_local, group=group)
    return _all_gather_padded(local, [int(c) for c in counts.tolist()], group)


def _all_gather_padded(local: torch.Tensor, n_per_rank: Sequence[int], group=None) -> List[torch.Tensor]:
    """One all_gather_into_tensor on a [W, cap, 64] buffer (cap = the largest shard's record count):
    a single fixed-size collective, no per-rank output copies."""
    world = dist.get_world_size(group)
    dev = _collective_device(local, group)
    cap = max(1, max(n_per_rank))
    padded = torch.zeros((cap, SAMPLE_BYTES), dtype=torch.uint8, device=dev)
    padded[: local.shape[0]] = local.to(dev)
    out = torch.empty((world, cap, SAMPLE_BYTES), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out.view(-1), padded.view(-1), group=group)   # flat views: the concatenated form every backend takes
    return [out[r, :c] for r, c in enumerate(n_per_rank)]


class ShardFailed(RuntimeError):
    """Another rank failed while playing its shard: raised on every rank instead of a hang in the exchange."""


def gather_shards(local_records: torch.Tensor, local_counts: np.ndarray, n_games: int, group=None, failed: bool = False
                  ) -> Tuple[List[torch.Tensor], List[np.ndarray]]:
    """The path's one exchange step.  `local_records` uint8[n_r, 64] = this rank's packed records in
    the order of its shard (`shard_indices`), `local_counts` the per-game sample counts of that shard.
    Two fixed-size collectives: the per-game counts (every rank knows the shard sizes, so no size
    exchange), then the records padded to the largest shard's record count (known from the counts).
    The counts message carries one status word: a rank whose play failed (`failed=True`) still takes
    part, and EVERY rank then raises `ShardFailed` instead of blocking in the second collective.
    Returns (records per rank, counts per rank) on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = _collective_device(local_records, group)
    cap_games = max(1, (n_games + world - 1) // world)
    mine = torch.zeros(cap_games + 1, dtype=torch.int32, device=dev)
    inconsistent = (not failed) and int(np.asarray(local_counts, dtype=np.int64).sum()) != local_records.shape[0]
    if not failed:
        mine[: len(local_counts)] = torch.as_tensor(np.asarray(local_counts).astype(np.int32), device=dev)
    mine[cap_games] = 1 if (failed or inconsistent) else 0
    all_counts = torch.empty((world, cap_games + 1), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(all_counts.view(-1), mine, group=group)
    all_counts = all_counts.cpu().numpy()
    bad = [r for r in range(world) if all_counts[r, cap_games] != 0]
    if inconsistent:
        raise ValueError(f"rank {rank}: {local_records.shape[0]} packed records but the counts add up to {int(np.asarray(local_counts, dtype=np.int64).sum())}")
    if bad:
        raise ShardFailed(f"self-play failed on rank(s) {bad}; no samples were exchanged")
    counts = [all_counts[r, : len(shard_indices(n_games, r, world))].astype(np.uint32) for r in range(world)]
    n_per_rank = [int(c.sum()) for c in counts]
    return _all_gather_padded(local_records, n_per_rank, group), counts


def merge_shards(per_rank_records: Sequence, per_rank_counts: Sequence[np.ndarray], n_games: int):
    """Interleave the ranks' packed records back into request-list order (vectorised; numpy
    SAMPLE_DTYPE arrays or torch uint8[n, 64] tensors).  Returns (records, counts[n_games])."""
    from .api import merge_parts

    world = len(per_rank_records)
    return merge_parts(n_games, [(shard_indices(n_games, r, world), per_rank_counts[r], per_rank_records[r]) for r in range(world)])


def play_games_sharded(reqs, max_nn_batch_size: int, n_mcts_iterations: int, c_exploration: float, c_ply_penalty: float,
                       *, evaluator, group=None, device=None, stats: Optional[dict] = None, **play_kwargs):
    """`play_games` over all ranks of an initialised process group (one process per GPU).

    Every rank passes the SAME `reqs`; rank r plays request positions r, r + W, ... in device mode on
    its own GPU with its own replica of `evaluator` (no traffic while games are played), packs its
    finished records on the device (k_pack_samples), takes part in the count + padded-record
    all-gathers (`gather_shards`; RCCL over xGMI with backend "nccl", host-staged with "gloo") and
    merges all shards: every rank returns the full `PlayGamesResult` in request order, identical to
    what a single process playing all of `reqs` returns.  `play_kwargs` are `play_games`' keywords
    (resident_games, concurrent_sessions, dirichlet, ...).  `stats` receives this rank's counters plus
    the exchange's wall time and byte count."""
    import time

    import inspect

    from .api import _ids_of, _play, _validate
    from .results import PlayGamesResult, results_from_records
    from .session import SAMPLE_DTYPE

    if not dist.is_initialized():
        raise RuntimeError("play_games_sharded needs torch.distributed.init_process_group() first")
    reqs = list(reqs)
    ids = _ids_of(reqs)
    _validate(reqs, max_nn_batch_size, n_mcts_iterations, None, evaluator, ids)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = len(reqs)
    if n == 0:
        return PlayGamesResult([])
    mine = ids[shard_indices(n, rank, world)]
    # every keyword `play_games` has for ONE device is a keyword here: the table is read off `_play`'s signature (what comes
    # after the reference's arguments and the two evaluator forms), so the two entry points cannot drift apart
    fixed = ("reqs", "max_nn_batch_size", "n_mcts_iterations", "c_exploration", "c_ply_penalty", "py_eval_pos_cb", "evaluator", "stats", "on_device")
    kw = {k: v.default for k, v in inspect.signature(_play).parameters.items() if k not in fixed}
    assert all(v is not inspect.Parameter.empty for v in kw.values())
    unknown = set(play_kwargs) - set(kw)
    if unknown:
        raise TypeError(f"unknown play_games keywords {sorted(unknown)}")
    kw.update(play_kwargs)
    if device is not None:
        kw["device"] = device
    if kw["device"] is None:
        # one process per GPU: LOCAL_RANK (set by torch.distributed.run) names this rank's device; without a
        # launcher, torch's current device -- never silently cuda:0 for every rank of a multi-GPU node
        import os
        lr = os.environ.get("LOCAL_RANK")
        if lr is not None and int(lr) < torch.cuda.device_count():
            kw["device"] = torch.device("cuda", int(lr))
        elif world > 1 and torch.cuda.device_count() > 1:
            raise ValueError("play_games_sharded: pass device= (or launch with torch.distributed.run, which sets LOCAL_RANK)")
        else:
            kw["device"] = torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(kw["device"])
    failure = None
    recs_dev, counts = torch.zeros((0, SAMPLE_BYTES), dtype=torch.uint8, device=dev), np.zeros(0, dtype=np.uint32)   # more ranks than games
    if len(mine):
        try:
            recs_dev, counts = _play(mine, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty, None, evaluator,
                                     stats=stats, on_device=True, **kw)
        except Exception as e:   # the other ranks are about to enter the exchange: tell them, then re-raise here
            failure = e
            recs_dev, counts = torch.zeros((0, SAMPLE_BYTES), dtype=torch.uint8, device=dev), np.zeros(0, dtype=np.uint32)
    t0 = time.perf_counter()
    try:
        per_rank, per_counts = gather_shards(recs_dev, counts, n, group, failed=failure is not None)
    except ShardFailed:
        if failure is not None:
            raise failure
        raise
    merged, all_counts = merge_shards(per_rank, per_counts, n)          # torch ops on the collective's device
    recs = merged.cpu().numpy().reshape(-1).view(SAMPLE_DTYPE)
    if stats is not None:
        stats["sample_allgather"] = {"ms": (time.perf_counter() - t0) * 1e3, "records_per_rank": [int(p.shape[0]) for p in per_rank],
                                     "bytes_total": int(sum(p.numel() for p in per_rank)), "backend": dist.get_backend(group)}
    return results_from_records(ids, recs, all_counts)
