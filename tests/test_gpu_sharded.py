"""SURVEY 8(e) with REAL sessions: two fresh processes share the one GPU of the test box, each
plays its shard through `c4a0_amd.distributed.play_games_sharded` (device-mode play, device packing,
count + padded-record all-gathers over gloo, vectorised merge) and every rank's merged result must
equal, byte for byte, what one process playing all requests returns -- and the oracle's samples."""
import os
import pickle
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("mode,n_games,n_iter", [("eager", 37, 12), ("graph2", 64, 20)])
def test_two_ranks_on_one_gpu_equal_the_single_process_result(tmp_path, mode, n_games, n_iter):
    world, port = 2, str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), str(r), str(world), port,
                               str(tmp_path), str(n_games), str(n_iter), mode], env=env, cwd=ROOT) for r in range(world)]
    # meanwhile: the same requests in this process, unsharded
    import torch
    from c4a0_amd import GameMetadata, play_games
    from oracle import c4oracle as O
    from tests.helpers import hash_eval_torch, oracle_samples_by_game

    reqs = [GameMetadata(1000 + 7 * i, 0, 0) for i in range(n_games)]
    single = play_games(reqs, 64, n_iter, 6.6, 0.01, evaluator=hash_eval_torch, device="cuda:0", resident_games=16)
    want = single.to_cbor()
    for p in procs:
        assert p.wait(timeout=600) == 0
    for r in range(world):
        got = pickle.load(open(tmp_path / f"rank{r}.pkl", "rb"))
        assert got["cbor"] == want, f"rank {r}: merged shards differ from the single-process result"
        ag = got["allgather"]
        assert ag["backend"] == "gloo" and sum(ag["records_per_rank"]) == sum(len(g.samples) for g in single.results)
    # and the single-process result is the oracle's
    ora, _ = O.self_play([(r.game_id, 0, 0) for r in reqs], 64, n_iter, 6.6, 0.01, "hash")
    ob = oracle_samples_by_game(ora)
    for g in single.results:
        mine = [(s.mask, s.value) for s in g.samples]
        assert mine == [(m, v) for m, v, *_ in ob[g.metadata.game_id]]


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path (ids sharded rank + W i, barriers, max-over-ranks timing, the sample
    exchange and its completeness check) with two ranks that share the one GPU: launched exactly as the
    driver launches it, except for the backend (gloo: RCCL refuses two ranks on one device)."""
    import json

    env = dict(os.environ, C4_BENCH_SAME_DEVICE="1", C4_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--rounds-per-step", "128", "--preroll", "1600"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "cpu_baseline" not in out                                  # rank 0 at N = 1 only
    ag = out["sample_allgather"]
    assert "error" not in ag, ag
    assert ag["merged_in_request_order_and_complete"] is True and ag["games_merged"] > 0
    assert len(ag["records_per_rank"]) == 2 and min(ag["records_per_rank"]) > 0
