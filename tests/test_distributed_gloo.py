"""The N>1 path on CPU: world_size-2 gloo run of the sharding + variable-length sample
all-gather used on 8 GPUs (c4a0_amd/distributed.py).  Each rank produces the records of its
shard (here with the oracle, standing in for the GPU session), all-gathers them and rebuilds the
request-order result; it must equal the single-process result game by game."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _records_for(ids):
    from c4a0_amd.session import SAMPLE_DTYPE
    from oracle import c4oracle as O

    res, _ = O.self_play([(g, 0, 0) for g in ids], 64, 6, 6.6, 0.01, "hash")
    recs, counts = [], []
    for g in ids:
        counts.append(len(res[g]))
        for i, s in enumerate(res[g]):
            r = np.zeros((), dtype=SAMPLE_DTYPE)
            r["game_id"], r["mask"], r["value"] = g, s.mask, s.value
            r["policy"], r["q_penalty"], r["q_no_penalty"] = s.policy, s.q_penalty, s.q_no_penalty
            r["meta"] = i | ((1 << 16) if i == len(res[g]) - 1 else 0)
            recs.append(r)
    return np.array(recs, dtype=SAMPLE_DTYPE), np.array(counts, dtype=np.uint32)


def _worker(rank, world, port, n_games, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from c4a0_amd.distributed import gather_shards, merge_shards, shard_indices
    from c4a0_amd.session import SAMPLE_DTYPE

    ids = np.arange(100, 100 + n_games)
    mine = ids[shard_indices(n_games, rank, world)]
    if len(mine):
        recs, counts = _records_for(mine.tolist())
        local = torch.from_numpy(recs.view(np.uint8).reshape(-1, 64).copy())
    else:   # more ranks than games: this rank owns nothing and still takes part in both collectives
        local, counts = torch.zeros((0, 64), dtype=torch.uint8), np.zeros(0, dtype=np.uint32)
    per_rank, per_counts = gather_shards(local, counts, n_games)          # the two collectives of the product path
    merged_t, mcounts = merge_shards(per_rank, per_counts, n_games)      # torch form (what play_games_sharded runs)
    merged = merged_t.numpy().reshape(-1).view(SAMPLE_DTYPE)
    merged_np, mcounts_np = merge_shards([p.numpy().reshape(-1).view(SAMPLE_DTYPE) for p in per_rank], per_counts, n_games)
    assert merged_np.tobytes() == merged.tobytes() and np.array_equal(mcounts, mcounts_np)   # numpy form agrees
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged)
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), mcounts)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_games", [1, 7, 10])
def test_two_rank_shard_and_allgather(tmp_path, n_games):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n_games, str(tmp_path)), nprocs=world, join=True)
    want_recs, want_counts = _records_for(list(range(100, 100 + n_games)))
    for rank in range(world):
        got = np.load(tmp_path / f"merged_{rank}.npy")
        cnt = np.load(tmp_path / f"counts_{rank}.npy")
        assert np.array_equal(cnt, want_counts)
        assert got.tobytes() == want_recs.tobytes()


def _failing_worker(rank, world, port, out_dir, n_games=6, bad_rank=1):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from c4a0_amd.distributed import ShardFailed, gather_shards, shard_indices

    mine = [100 + int(i) for i in shard_indices(n_games, rank, world)]
    if mine:
        recs, counts = _records_for(mine)
        local = torch.from_numpy(recs.view(np.uint8).reshape(-1, 64).copy())
    else:   # more ranks than games: an empty shard still takes part in both collectives
        local, counts = torch.zeros((0, 64), dtype=torch.uint8), np.zeros(0, dtype=np.uint32)
    try:
        if rank == bad_rank:   # this rank's play "failed": it still joins the counts exchange, flagged
            gather_shards(torch.zeros((0, 64), dtype=torch.uint8), np.zeros(0, dtype=np.uint32), n_games, failed=True)
        else:
            gather_shards(local, counts, n_games)
        outcome = "returned"
    except ShardFailed as e:
        outcome = f"ShardFailed: {e}"
    open(os.path.join(out_dir, f"outcome_{rank}.txt"), "w").write(outcome)
    dist.barrier()
    dist.destroy_process_group()


def test_a_failed_rank_makes_every_rank_raise_instead_of_hanging(tmp_path):
    """without agreement a rank that raised in _play left the others blocked in all_gather."""
    world, port = 2, _free_port()
    mp.spawn(_failing_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        out = open(tmp_path / f"outcome_{rank}.txt").read()
        assert out.startswith("ShardFailed") and "[1]" in out, out


@pytest.mark.parametrize("n_games", [5, 13, 64])
def test_eight_rank_shard_and_allgather(tmp_path, n_games):
    """BASELINE configs 3 / 5 run on 8 ranks: the exchange at world size 8 with ranks that own NO game (5 games), uneven
    shards (13 = 5 ranks with two games, 3 with one) and even ones (64).  Every rank must rebuild the single-process result."""
    world, port = 8, _free_port()
    mp.spawn(_worker, args=(world, port, n_games, str(tmp_path)), nprocs=world, join=True)
    want_recs, want_counts = _records_for(list(range(100, 100 + n_games)))
    for rank in range(world):
        got = np.load(tmp_path / f"merged_{rank}.npy")
        cnt = np.load(tmp_path / f"counts_{rank}.npy")
        assert np.array_equal(cnt, want_counts), rank
        assert got.tobytes() == want_recs.tobytes(), rank


@pytest.mark.parametrize("n_games,bad_rank", [(13, 5), (5, 7)])
def test_eight_ranks_one_failing_rank_raises_everywhere(tmp_path, n_games, bad_rank):
    """World size 8, uneven shards; the failing rank owns games (13 games, rank 5) or none at all (5 games, rank 7)."""
    world, port = 8, _free_port()
    mp.spawn(_failing_worker, args=(world, port, str(tmp_path), n_games, bad_rank), nprocs=world, join=True)
    for rank in range(world):
        out = open(tmp_path / f"outcome_{rank}.txt").read()
        assert out.startswith("ShardFailed") and f"[{bad_rank}]" in out, (rank, out)


def test_shard_indices_partition():
    from c4a0_amd.distributed import shard_indices

    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            parts = [shard_indices(n, r, w) for r in range(w)]
            assert sorted(np.concatenate(parts).tolist()) == list(range(n))
