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
