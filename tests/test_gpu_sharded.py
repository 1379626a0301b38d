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
    from tests.helpers import hash_eval_torch

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
    _assert_equals_oracle(single, reqs, n_iter)


def _assert_equals_oracle(result, reqs, n_iter):
    """every sample of every game -- position, policy bits, both q bits -- is the oracle's"""
    import numpy as np
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game

    ora, _ = O.self_play([(r.game_id, 0, 0) for r in reqs], 64, n_iter, 6.6, 0.01, "hash")
    ob = oracle_samples_by_game(ora)
    assert len(result.results) == len(reqs)
    for g in result.results:
        mine = [(s.mask, s.value, np.asarray(s.policy, dtype=np.float32).tobytes(), np.float32(s.q_penalty).tobytes(),
                 np.float32(s.q_no_penalty).tobytes()) for s in g.samples]
        assert mine == ob[g.metadata.game_id], f"game {g.metadata.game_id}"


@pytest.mark.parametrize("mode,n_games,n_iter", [("eager", 37, 12), ("graph2", 64, 20)])
def test_rccl_world_size_one_equals_play_games(tmp_path, mode, n_games, n_iter):
    """The product path over the REAL backend ("nccl" = RCCL): device pack -> RCCL all_gather_into_tensor
    of counts and records -> device merge.  RCCL takes one rank per device, so on the one-GPU box the
    world has one rank; the collectives, buffers and merge are the ones an 8-GPU job runs."""
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), "0", "1", port, str(tmp_path),
                          str(n_games), str(n_iter), mode, "nccl"], env=env, cwd=ROOT)
    from c4a0_amd import GameMetadata, play_games
    from tests.helpers import hash_eval_torch

    reqs = [GameMetadata(1000 + 7 * i, 0, 0) for i in range(n_games)]
    single = play_games(reqs, 64, n_iter, 6.6, 0.01, evaluator=hash_eval_torch, device="cuda:0", resident_games=16)
    assert p.wait(timeout=600) == 0
    got = pickle.load(open(tmp_path / "rank0.pkl", "rb"))
    assert got["allgather"]["backend"] == "nccl"
    assert got["cbor"] == single.to_cbor()
    _assert_equals_oracle(single, reqs, n_iter)


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_sharded_play_with_reclaimed_arenas_beyond_the_old_search_limit(tmp_path, backend, world):
    """`play_games_sharded` takes every keyword `play_games` takes (its table is read off `_play`'s signature),
    here reclaim / reclaim_period / blocks_per_slot: n_mcts_iterations = 1 600 (the never-reclaimed arena stops at 1 523) with the
    tightest halves the library accepts, two ranks over gloo and one over RCCL; the merged result is the oracle's, sample for
    sample, and the arenas really were reclaimed on every rank."""
    n_games, n_iter, port = 6, 1600, str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), str(r), str(world), port,
                               str(tmp_path), str(n_games), str(n_iter), "reclaim", backend], env=env, cwd=ROOT) for r in range(world)]
    import numpy as np
    from c4a0_amd import GameMetadata, PlayGamesResult
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game

    reqs = [GameMetadata(1000 + 7 * i, 0, 0) for i in range(n_games)]
    ora, _ = O.self_play([(r.game_id, 0, 0) for r in reqs], 64, n_iter, 6.6, 0.01, "hash")
    want = oracle_samples_by_game(ora)
    for p in procs:
        assert p.wait(timeout=900) == 0
    for r in range(world):
        got = pickle.load(open(tmp_path / f"rank{r}.pkl", "rb"))
        assert got["allgather"]["backend"] == backend and got["reclaim_passes"] > n_games // world, got["reclaim_passes"]
        res = PlayGamesResult.from_cbor(got["cbor"])
        assert [g.metadata.game_id for g in res.results] == [q.game_id for q in reqs]
        for g in res.results:
            mine = [(s.mask, s.value, np.asarray(s.policy, dtype=np.float32).tobytes(), np.float32(s.q_penalty).tobytes(),
                     np.float32(s.q_no_penalty).tobytes()) for s in g.samples]
            assert mine == want[g.metadata.game_id], f"rank {r} game {g.metadata.game_id}"


def test_sharded_keywords_are_play_games_keywords():
    """The keyword table of play_games_sharded IS _play's signature: every keyword-only argument of play_games (except the
    evaluator forms and stats, which the sharded entry point has itself) is accepted, anything else is refused by name."""
    import inspect
    from c4a0_amd.api import _play, play_games

    play_kw = {k for k, v in inspect.signature(play_games).parameters.items() if v.kind is inspect.Parameter.KEYWORD_ONLY} - {"evaluator", "stats"}
    own = set(inspect.signature(_play).parameters) - {"reqs", "max_nn_batch_size", "n_mcts_iterations", "c_exploration", "c_ply_penalty",
                                                      "py_eval_pos_cb", "evaluator", "stats", "on_device"}
    assert play_kw == own, play_kw ^ own


def _check_bench_line(r, n_gpus):
    import json

    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n_gpus and out["scaling"] == "weak" and out["value"] > 0
    ag = out["sample_allgather"]
    assert "error" not in ag, ag
    assert ag["merged_in_request_order_and_complete"] is True and ag["games_merged"] > 0 and ag["ms"] > 0
    assert len(ag["records_per_rank"]) == n_gpus and min(ag["records_per_rank"]) > 0
    assert len(out["per_rank"]["games_completed"]) == n_gpus and len(out["per_rank"]["elapsed_s"]) == n_gpus
    return out


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher: bench.py starts the two rank processes itself (fresh
    children, before any GPU call), relays the one JSON line and the exit code.  Two ranks share the one
    GPU of the test box, hence gloo."""
    env = dict(os.environ, C4_BENCH_SAME_DEVICE="1", C4_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--rounds-per-step", "128",
           "--preroll", "1600"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = _check_bench_line(r, 2)
    assert "cpu_baseline" not in out


def test_bench_at_world_size_eight_on_one_gpu(tmp_path):
    """BASELINE configs 3 / 5 run 8 ranks.  No 8-GPU node is available to the tests, so: `bench.py --gpus 8` exactly as the driver
    calls it at N = 8 (it starts its own 8 rank processes), all eight on the test box's one GPU over gloo, at a small per-rank size.
    What must hold is everything but the speed: 8 entries per rank-indexed field, every rank's records present, and the merged
    result complete and in request order."""
    env = dict(os.environ, C4_BENCH_SAME_DEVICE="1", C4_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "1", "--rounds-per-step", "128",
           "--preroll", "1600", "--games-per-gpu", "512"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    out = _check_bench_line(r, 8)
    assert "cpu_baseline" not in out and out["config"]["parallelism"] == "games sharded id%8"
    assert len(out["per_rank"]["games_completed"]) == 8 and all(g > 0 for g in out["per_rank"]["games_completed"])


def test_bench_over_rccl_at_world_size_one(tmp_path):
    """bench.py's N > 1 branch (barriers, reductions, the sample exchange and its check) over backend
    "nccl" = RCCL, forced on at world size 1 (C4_BENCH_FORCE_DIST): what the 8-GPU node will run, minus peers."""
    env = dict(os.environ, C4_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_PORT=str(_free_port()))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "C4_BENCH_BACKEND", "C4_BENCH_SAME_DEVICE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--rounds-per-step", "128",
           "--preroll", "1600", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    _check_bench_line(r, 1)


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 code path (ids sharded rank + W i, barriers, max-over-ranks timing, the sample
    exchange and its completeness check) launched exactly as the driver launches it (torch.distributed.run),
    except for the backend (gloo: RCCL refuses two ranks on one device)."""
    env = dict(os.environ, C4_BENCH_SAME_DEVICE="1", C4_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--rounds-per-step", "128", "--preroll", "1600"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = _check_bench_line(r, 2)
    assert "cpu_baseline" not in out                                  # rank 0 at N = 1 only


def test_rccl_live_during_graph_capture_collective_before_play_and_two_calls(tmp_path):
    """the way a training loop uses the sharded entry point.  The RCCL process group is up
    and has just run collectives (a broadcast of every weight tensor from rank 0, a barrier) when
    `play_games_sharded` captures its HIP graphs (two sessions, the paired graph of session.capture_pair, with
    the bf16 network); another collective follows; then `play_games_sharded` is called AGAIN in the same
    process (fresh sessions, fresh capture, RCCL's watchdog thread polling its events all along).  Both calls
    must return the single-process result byte for byte."""
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    n_games, n_iter = 96, 12
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharded_worker.py"), "0", "1", port, str(tmp_path),
                          str(n_games), str(n_iter), "rccl_live", "nccl"], env=env, cwd=ROOT)
    import torch
    from c4a0_amd import GameMetadata, play_games
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), torch.device("cuda:0"), dtype=torch.bfloat16)
    reqs = [GameMetadata(1000 + 7 * i, 0, 0) for i in range(n_games)]
    single = play_games(reqs, 64, n_iter, 6.6, 0.01, evaluator=net, device="cuda:0", resident_games=16, concurrent_sessions=1)
    assert p.wait(timeout=600) == 0
    got = pickle.load(open(tmp_path / "rank0.pkl", "rb"))
    assert got["allgather"]["backend"] == "nccl" and got["collectives_before_play"] >= 2
    assert got["host_loop"] == "native"      # the rank's rounds ran inside c4_play_games_bf16 (its graph captured with RCCL's watchdog thread alive), records left on the device
    assert got["cbor"] == single.to_cbor(), "first call (right after a broadcast + barrier) differs"
    assert got["cbor_second_call"] == single.to_cbor(), "second call in the same process differs"
