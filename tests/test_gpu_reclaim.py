"""Reclaimed tree arenas (C4_FLAG_RECLAIM, k_arena_reclaim): the reference's tree is heap nodes freed at every re-root
(rust/src/mcts.rs:187-206, 332-355) and takes any n_mcts_iterations; the never-reclaimed arena of rounds 1-4 refused n > 1 523.
A reclaimed arena copies the live subtree into its other half when one half runs short.  Which block a node sits in changes nothing
a game records: every sample must equal the oracle's (and the never-reclaimed arena's), whatever the period, the size of the halves,
the launch form (eager, HIP graph, paired sessions, numpy callback) or the extensions in use."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _half_min(n, period, max_sims=2):
    return n + max_sims + 8 + 2 * (2 * period * max_sims + 16)


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from oracle import c4oracle as O
    return O, torch.device("cuda:0")


def test_search_width_beyond_the_old_limit_equals_the_oracle(env):
    """play_games(n_mcts_iterations=5000) on two games: accepted with the default sizing, reclaimed on the way,
    every sample identical to the oracle's."""
    import c4a0_amd
    from tests.helpers import hash_eval_torch, oracle_samples_by_game
    O, dev = env
    reqs = [(7, 0, 0), (2 ** 40 + 3, 0, 0)]
    stats = {}
    got = c4a0_amd.play_games([c4a0_amd.GameMetadata(*r) for r in reqs], 64, 5000, 6.6, 0.01, evaluator=hash_eval_torch, device=dev, stats=stats)
    want, _ = O.self_play(reqs, 64, 5000, 6.6, 0.01, "hash")
    by_game = {r.metadata.game_id: [(x.mask, x.value, x.policy.tobytes(), x.q_penalty.tobytes(), x.q_no_penalty.tobytes()) for x in r.samples]
               for r in got.results}
    assert by_game == oracle_samples_by_game(want)
    assert stats["error"] == 0 and stats["games_done"] == 2
    assert stats["reclaim_passes"] >= 2 and stats["reclaim_blocks"] >= stats["reclaim_passes"], stats


def test_default_arena_of_the_references_job_is_small_and_the_limits_are_stated(env):
    """1 700 slots at n = 1 400 (src/c4a0/main.py:40-51): <= 2 GB of arena (13 GB never reclaimed); BASELINE's shapes (n <= 800) keep
    the never-reclaimed arena; what is still refused is refused with the reason."""
    from c4a0_amd._lib import C4Error
    from c4a0_amd.session import DeviceSession
    O, dev = env
    s = DeviceSession(1700, 1400, 6.6, 0.01, device=dev)
    a = s.arena()
    s.close()
    assert a["reclaim_half_blocks"] > 0 and a["bytes"] <= 2 * 10 ** 9, a
    for n in (100, 800, 1000):
        s = DeviceSession(4, n, 6.6, 0.01, device=dev)
        a = s.arena()
        s.close()
        assert a["reclaim_half_blocks"] == 0 and a["blocks_per_slot"] == 43 * n + 8, (n, a)
    s = DeviceSession(2, 1400, 6.6, 0.01, device=dev, reclaim=False)           # the never-reclaimed arena on request
    assert s.arena() == {"bytes": 2 * (43 * 1400 + 8) * 128, "blocks_per_slot": 43 * 1400 + 8, "reclaim_half_blocks": 0}
    s.close()
    s = DeviceSession(2, 30000, 6.6, 0.01, device=dev)                          # 16-bit links reach this far now
    assert s.arena()["reclaim_half_blocks"] == 32767
    s.close()
    with pytest.raises(C4Error, match="reclaimed arena too small"):
        DeviceSession(2, 40000, 6.6, 0.01, device=dev)
    with pytest.raises(C4Error, match="n_mcts_iterations > 1523"):
        DeviceSession(2, 2000, 6.6, 0.01, device=dev, reclaim=False)
    with pytest.raises(C4Error, match="n_mcts_iterations > 1523"):
        DeviceSession(2, 2000, 6.6, 0.01, device=dev, no_moves=True)           # a search that never moves never frees anything
    with pytest.raises(C4Error, match="reclaimed arena too small"):
        DeviceSession(2, 50, 6.6, 0.01, device=dev, reclaim=True, blocks_per_slot=2 * (_half_min(50, 64) - 1))
    s = DeviceSession(2, 50, 6.6, 0.01, device=dev, reclaim=True, blocks_per_slot=2 * _half_min(50, 64))
    s.close()


@pytest.mark.parametrize("period,extra,graph", [(1, 0, 0), (1, 3, 4), (2, 0, 1), (3, 10, 8), (5, 0, 0), (16, 40, 64)])
@pytest.mark.parametrize("planes_dtype", ["f32", "bf16"])
def test_tight_halves_reclaimed_all_the_time_equal_the_oracle(env, period, extra, graph, planes_dtype):
    """Halves as small as the library accepts and a look at the arenas every `period`-th launch: a game is compacted several times per
    move.  Eager steps and HIP graphs of every length around the period (each capture restarts the count), refill of finished games,
    start positions, 64-bit ids."""
    from c4a0_amd.session import DeviceSession
    from tests.helpers import GraphSafeHashEval, hash_eval_torch, oracle_samples_by_game, samples_by_game
    O, dev = env
    n = 14
    reqs = [(g, 0, 0) for g in [0, 1, 42, 2 ** 64 - 1, 2 ** 63 + 5] + list(range(1000, 1040))]
    s = DeviceSession(9, n, 6.6, 0.01, device=dev, planes_dtype=torch.float32 if planes_dtype == "f32" else torch.bfloat16,
                      reclaim=True, reclaim_period=period, blocks_per_slot=2 * (_half_min(n, period) + extra))
    s.set_games(reqs)
    if graph:
        s.run(GraphSafeHashEval(), steps_per_graph=graph)
    else:
        s.run(hash_eval_torch)
    got = samples_by_game(s.drain_samples())
    c = s.counters()
    s.close()
    want, _ = O.self_play(reqs, 64, n, 6.6, 0.01, "hash")
    assert c["error"] == 0 and c["games_done"] == len(reqs)
    assert got == oracle_samples_by_game(want)
    # really reclaimed: with the tightest halves several times per game, with a long period (large halves: 4 x period launches' blocks) at least now and then
    assert c["reclaim_passes"] > (2 * len(reqs) if period == 1 else (len(reqs) if period <= 3 else 3)), c


@pytest.mark.parametrize("ext", ["dirichlet", "cache", "one_sim"])
def test_reclaim_with_the_extensions(env, ext):
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game
    O, dev = env
    n = 20
    reqs = [(300 + 3 * i, 0, 0) for i in range(30)]
    s = DeviceSession(8, n, 1.4, 0.01, device=dev, reclaim=True, reclaim_period=2, blocks_per_slot=2 * (_half_min(n, 2, 8) + 5),
                      one_sim_per_step=ext == "one_sim")
    s.set_games(reqs)
    if ext == "dirichlet":
        s.set_dirichlet(0.3, 0.25)
    if ext == "cache":
        s.set_eval_cache(4096)
    s.run(hash_eval_torch)
    got = samples_by_game(s.drain_samples())
    c = s.counters()
    s.close()
    want, _ = O.self_play(reqs, 64, n, 1.4, 0.01, "hash", dirichlet=(0.3, 0.25) if ext == "dirichlet" else (0.0, 0.0))
    assert c["error"] == 0 and got == oracle_samples_by_game(want)
    assert c["reclaim_passes"] > len(reqs) // 2, c


def test_reclaimed_and_never_reclaimed_arenas_record_the_same_bytes_with_the_network(env):
    """The bf16 network, two paired sessions in one HIP graph (the fused output + step launch), tail narrowing (games move between
    slots, arenas and all), and the numpy-callback mode: byte-identical records with the arenas reclaimed every fourth launch."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    O, dev = env
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    reqs = [c4a0_amd.GameMetadata(900 + i, 0, 0) for i in range(700)]
    n = 24
    kw = dict(resident_games=600, concurrent_sessions=2)
    base = c4a0_amd.play_games(reqs, 64, n, 6.6, 0.01, evaluator=net, device=dev, reclaim=False, **kw)
    st = {}
    got = c4a0_amd.play_games(reqs, 64, n, 6.6, 0.01, evaluator=net, device=dev, reclaim=True, reclaim_period=4,
                              blocks_per_slot=2 * (_half_min(n, 4) + 6), stats=st, **kw)
    assert st["reclaim_passes"] > 700 and st["rows_at_end"] < st["n_slots"]          # reclaimed, and narrowed at the tail
    assert got.to_records()[0].tobytes() == base.to_records()[0].tobytes()
    st = {}
    cb = c4a0_amd.play_games(reqs[:90], 2000, n, 6.6, 0.01, lambda _m, x: net.forward_numpy(x), device=dev, resident_games=64, reclaim=True,
                             reclaim_period=4, blocks_per_slot=2 * (_half_min(n, 4) + 6), stats=st)
    base90 = c4a0_amd.play_games(reqs[:90], 64, n, 6.6, 0.01, evaluator=net, device=dev, reclaim=False, resident_games=64, concurrent_sessions=1)
    assert st["reclaim_passes"] > 90
    assert cb.to_records()[0].tobytes() == base90.to_records()[0].tobytes()
