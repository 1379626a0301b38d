"""GPU tests of the drop-in entry point `play_games` (reference rust/src/pybridge.rs:20-53), written
after the reference's own boundary tests (tests/c4a0_tests/pybridge_test.py, tournament_test.py)
plus parity of the results against the CPU oracle."""
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _uniform_eval(_model_id, pos):                      # pybridge_test.py:7-11
    b = pos.shape[0]
    return np.zeros((b, 7), dtype=np.float32), np.zeros((b,), dtype=np.float32), np.zeros((b,), dtype=np.float32)


def _as_oracle_dict(result):
    return {r.metadata.game_id: [(s.mask, s.value, s.policy.tobytes(), s.q_penalty.tobytes(), s.q_no_penalty.tobytes())
                                 for s in r.samples] for r in result.results}


def test_split_train_test_is_deterministic_and_non_mutating():
    """pybridge_test.py:22-39, verbatim scenario through the GPU-backed play_games."""
    import c4a0_amd as c4a0_rust

    games = c4a0_rust.play_games([c4a0_rust.GameMetadata(i, 0, 0) for i in range(4)], 8, 2, 1.4, 0.01, _uniform_eval)
    ids = [r.metadata.game_id for r in games.results]
    first_train, first_test = games.split_train_test(0.5, 1337)
    assert [r.metadata.game_id for r in games.results] == ids
    second_train, second_test = games.split_train_test(0.5, 1337)
    assert [s.pos_str() for s in first_train] == [s.pos_str() for s in second_train]
    assert [s.pos_str() for s in first_test] == [s.pos_str() for s in second_test]


def test_callback_mode_matches_oracle_and_honours_the_callback_contract():
    """BASELINE config 1 shape (32 games, n_mcts_iterations = 10) through the numpy callback:
    batches <= max_nn_batch_size, unique positions, float32 [B,2,6,7] (pybridge.rs:161-221),
    samples identical to the oracle driven by the same callback."""
    import c4a0_amd
    from oracle import c4oracle as O
    from tests.helpers import hash_eval_np, oracle_samples_by_game

    seen = []

    def cb(model_id, x):
        assert x.dtype == np.float32 and x.shape[1:] == (2, 6, 7) and x.flags["C_CONTIGUOUS"]
        assert len({x[i].tobytes() for i in range(x.shape[0])}) == x.shape[0]
        seen.append(x.shape[0])
        return hash_eval_np(model_id, x)

    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(32)]
    stats = {}
    got = c4a0_amd.play_games(reqs, 5, 10, 6.6, 0.01, cb, stats=stats)
    assert max(seen) <= 5 and seen[0] == 1                  # all games start on the same position
    want, ost = O.self_play([(i, 0, 0) for i in range(32)], 5, 10, 6.6, 0.01, "hash")
    assert _as_oracle_dict(got) == oracle_samples_by_game(want)
    assert [r.metadata.game_id for r in got.results] == list(range(32))
    assert stats["games_done"] == 32 and stats["samples"] == ost["n_samples"]
    # the result container round-trips through CBOR and pickle (pybridge.rs:73-92)
    assert type(got).from_cbor(got.to_cbor()) == got and pickle.loads(pickle.dumps(got)) == got
    # every game ends in exactly one terminal sample whose score is defined (types.rs:77-99)
    assert all(0.0 <= r.player0_score() <= 1.0 for r in got.results)


def test_callback_errors_surface_as_exceptions():
    import c4a0_amd

    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(3)]
    with pytest.raises(TypeError):      # wrong dtype (reference: "Failed to extract result", pybridge.rs:184)
        c4a0_amd.play_games(reqs, 8, 2, 1.4, 0.01, lambda m, x: (np.zeros((x.shape[0], 7)), np.zeros(x.shape[0]), np.zeros(x.shape[0])))
    with pytest.raises(ZeroDivisionError):   # callback raised (reference: panic, pybridge.rs:182-183)
        c4a0_amd.play_games(reqs, 8, 2, 1.4, 0.01, lambda m, x: 1 / 0)
    with pytest.raises(TypeError):
        c4a0_amd.play_games(reqs, 8, 2, 1.4, 0.01)
    assert c4a0_amd.play_games([], 8, 2, 1.4, 0.01, _uniform_eval).results == []


def test_tournament_style_multi_model_games():
    """tournament_test.py:27-51: players with different ids; the callback is dispatched on the
    model to play at the leaf (mcts.rs:70-76); ids are preserved and scores are in [0, 1].
    Per-game results equal the oracle's (which restates the reference's majority-model batching)."""
    import itertools

    import c4a0_amd
    from oracle import c4oracle as O
    from tests.helpers import hash_eval_np, oracle_samples_by_game

    def player(model_id, x):
        lp, qp, qn = hash_eval_np(model_id, x)
        return np.ascontiguousarray(np.roll(lp, int(model_id), axis=1)), qp, qn   # each "model" prefers other columns

    calls = []

    def cb(model_id, x):
        calls.append(model_id)
        return player(model_id, x)

    pairings = list(itertools.permutations([0, 1, 2], 2))
    reqs = [c4a0_amd.GameMetadata(i, p0, p1) for i, (p0, p1) in enumerate(pairings)]
    got = c4a0_amd.play_games(reqs, 4, 6, 1.4, 0.01, cb)
    assert set(calls) == {0, 1, 2}
    assert [(r.metadata.game_id, r.metadata.player0_id, r.metadata.player1_id) for r in got.results] == \
           [(i, p0, p1) for i, (p0, p1) in enumerate(pairings)]
    assert all(0.0 <= r.player0_score() <= 1.0 for r in got.results)
    want, _ = O.self_play([(i, p0, p1) for i, (p0, p1) in enumerate(pairings)], 4, 6, 1.4, 0.01, player)
    assert _as_oracle_dict(got) == oracle_samples_by_game(want)


@pytest.mark.parametrize("blocks,channels,n_iter", [
    (1, 32, 12),    # small network, shallow search
    (2, 64, 40),    # the 64-channel tower (BASELINE configs 4/5 family) and the F = 2688 head kernel, deeper trees
])
def test_device_mode_with_real_network_and_t3_replay_parity(blocks, channels, n_iter):
    """Device mode (leaves never leave HBM) with the bf16 ResNet, HIP-graph replayed.  T3 parity
    (SURVEY 8c): the run logs every (leaf position -> evaluator output) pair it consumed; the
    oracle replays the games with a lookup evaluator and must emit identical samples."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from c4a0_amd.session import DeviceSession
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game, planes_to_pos_np, samples_by_game

    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(blocks, channels, 2, 2)), dev, dtype=torch.bfloat16)
    reqs = [(i, 0, 0) for i in range(24)]
    s = DeviceSession(16, n_iter, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games(reqs)
    table = {}

    def log(_step):
        # at this point planes hold the leaves and logprobs/q the evaluator's answers for them
        _m, _v, status = s.leaves()
        mask, value = planes_to_pos_np(s.planes.float().cpu().numpy())
        lp, q = s.logprobs.cpu().numpy(), s.q.cpu().numpy()
        for g in np.nonzero(status == 1)[0]:
            key = (int(mask[g]), int(value[g]))
            val = (lp[g].tobytes(), q[g].tobytes())
            assert table.setdefault(key, val) == val, "evaluator must be a function of the position"
    s.run(net, on_step=log)
    got = samples_by_game(s.drain_samples())
    s.close()

    zeros = (np.zeros(7, np.float32).tobytes(), np.zeros(2, np.float32).tobytes())

    def answer(m, v):
        # the device never shows a terminal leaf to the evaluator (its simulation runs in the launch
        # that selected it); the reference asks and ignores the answer (mcts.rs:92-98)
        if (m, v) not in table:
            assert O.terminal_state(O.Pos(m, v)) != 0, "a non-terminal leaf the device never evaluated"
            return zeros
        return table[(m, v)]

    def lookup(_model_id, x):
        mask, value = planes_to_pos_np(x)
        ans = [answer(int(m), int(v)) for m, v in zip(mask, value)]
        lp = np.stack([np.frombuffer(a[0], dtype=np.float32) for a in ans])
        q = np.stack([np.frombuffer(a[1], dtype=np.float32) for a in ans])
        return np.ascontiguousarray(lp), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])

    want, _ = O.self_play(reqs, 64, n_iter, 6.6, 0.01, lookup)
    assert got == oracle_samples_by_game(want)

    # the same run through play_games(evaluator=...), HIP-graph replayed: identical samples again
    res = c4a0_amd.play_games([c4a0_amd.GameMetadata(*r) for r in reqs], 64, n_iter, 6.6, 0.01, evaluator=net,
                              resident_games=16, planes_dtype=torch.bfloat16)
    assert _as_oracle_dict(res) == oracle_samples_by_game(want)


def test_device_mode_tournament_matches_callback_mode_and_oracle():
    """Games between different models with DEVICE evaluators (evaluator={model_id: callable}):
    the step kernel routes every leaf to the model to play (mcts.rs:70-76).  Same samples as the
    oracle's restatement of the reference scheduler with the numpy twins of the evaluators."""
    import itertools

    import c4a0_amd
    from oracle import c4oracle as O
    from tests.helpers import hash_eval_np, hash_eval_torch, oracle_samples_by_game

    def dev_player(mid):
        def f(planes):
            lp, q = hash_eval_torch(planes)
            return torch.roll(lp, mid, dims=1), q
        return f

    def np_player(model_id, x):
        lp, qp, qn = hash_eval_np(model_id, x)
        return np.ascontiguousarray(np.roll(lp, int(model_id), axis=1)), qp, qn

    pairings = list(itertools.permutations([3, 5, 9], 2)) * 2
    reqs = [c4a0_amd.GameMetadata(100 + i, p0, p1) for i, (p0, p1) in enumerate(pairings)]
    seen = {3: 0, 5: 0, 9: 0}

    def counting(mid):
        f = dev_player(mid)

        def g(planes):
            seen[mid] += planes.shape[0]          # every model is handed its own rows only
            return f(planes)
        return g

    stats = {}
    got = c4a0_amd.play_games(reqs, 64, 8, 1.4, 0.01, evaluator={m: counting(m) for m in (3, 5, 9)}, resident_games=8, stats=stats)
    want, _ = O.self_play([(r.game_id, r.player0_id, r.player1_id) for r in reqs], 64, 8, 1.4, 0.01, np_player)
    assert _as_oracle_dict(got) == oracle_samples_by_game(want)
    assert all(0.0 <= r.player0_score() <= 1.0 for r in got.results)
    assert 0 < sum(seen.values()) <= 8 * stats["steps"] and all(seen.values())
    # 64-bit model ids: patterns >= 2^63 are routed like any other (they are negative in the device's int64 tensor)
    big = (1 << 63) + 7
    reqs_big = [c4a0_amd.GameMetadata(200 + i, *(pair if i % 2 else pair[::-1])) for i, pair in enumerate([(4, big)] * 6)]
    got = c4a0_amd.play_games(reqs_big, 64, 8, 1.4, 0.01, evaluator={4: dev_player(4), big: dev_player(big % 7)}, resident_games=4)
    want, _ = O.self_play([(r.game_id, r.player0_id, r.player1_id) for r in reqs_big], 64, 8, 1.4, 0.01,
                          lambda m, x: np_player(m if m == 4 else big % 7, x))
    assert _as_oracle_dict(got) == oracle_samples_by_game(want)
    with pytest.raises(KeyError):
        c4a0_amd.play_games(reqs, 64, 8, 1.4, 0.01, evaluator={3: dev_player(3)})
    with pytest.raises(TypeError):
        c4a0_amd.play_games(reqs, 64, 8, 1.4, 0.01, evaluator=dev_player(3))


def test_concurrent_sessions_give_the_same_samples_as_one_session():
    """play_games(concurrent_sessions=2): the resident games are split over two sessions replaying
    their HIP graphs on separate streams (session.run_sessions).  Same samples, same order, as one
    session; also with more games than slots (refill) and an odd split."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    reqs = [c4a0_amd.GameMetadata(100 + i, 0, 0) for i in range(37)]
    st1, st2, st3 = {}, {}, {}
    one = c4a0_amd.play_games(reqs, 64, 9, 6.6, 0.01, evaluator=net, resident_games=16, concurrent_sessions=1, stats=st1)
    two = c4a0_amd.play_games(reqs, 64, 9, 6.6, 0.01, evaluator=net, resident_games=16, concurrent_sessions=2, stats=st2)
    three = c4a0_amd.play_games(reqs, 64, 9, 6.6, 0.01, evaluator=net, resident_games=15, concurrent_sessions=3, stats=st3)
    assert st1["concurrent_sessions"] == 1 and st2["concurrent_sessions"] == 2 and st3["concurrent_sessions"] == 3
    assert st2["n_slots"] == 16 and st3["n_slots"] == 15
    a = one.to_arrays()
    for other, st in ((two, st2), (three, st3)):
        b = other.to_arrays()
        assert [r.metadata.game_id for r in other.results] == [r.game_id for r in reqs]
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
        for k in ("sims", "moves", "games_done", "samples", "expansions", "backup_nodes"):
            assert st[k] == st1[k], k


def test_evaluation_cache_with_the_real_network_changes_no_sample():
    """play_games(eval_cache_entries=...) with the bf16 network, one and two concurrent sessions: the
    cached outputs are the network's own earlier answers for the same position, so every sample
    equals the run without the cache, and the evaluator passes drop."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    reqs = [c4a0_amd.GameMetadata(500 + i, 0, 0) for i in range(48)]
    st0, st1, st2 = {}, {}, {}
    base = c4a0_amd.play_games(reqs, 64, 25, 6.6, 0.01, evaluator=net, resident_games=32, concurrent_sessions=1, stats=st0)
    one = c4a0_amd.play_games(reqs, 64, 25, 6.6, 0.01, evaluator=net, resident_games=32, concurrent_sessions=1,
                              eval_cache_entries=1 << 16, stats=st1)
    two = c4a0_amd.play_games(reqs, 64, 25, 6.6, 0.01, evaluator=net, resident_games=32, concurrent_sessions=2,
                              eval_cache_entries=1 << 16, stats=st2)
    a = base.to_arrays()
    for other in (one, two):
        for x, y in zip(a, other.to_arrays()):
            assert np.array_equal(x, y)
    assert st0["eval_cache_hits"] == 0 and st1["eval_cache_hits"] > 0 and st2["eval_cache_hits"] > 0
    assert st1["sims"] == st0["sims"] == st2["sims"] and st1["steps"] < st0["steps"]


def test_device_resident_training_tensors_equal_the_reference_conversion():
    """finished samples -> training tensors without leaving HBM
    (c4a0_amd.dataset).  Must equal `Sample.to_numpy()` (types.rs:125-147) of every sample followed,
    like the reference's SampleDataModule (training.py:317), by every sample's `flip_h()`."""
    from c4a0_amd import GameMetadata
    from c4a0_amd.dataset import training_tensors
    from c4a0_amd.results import results_from_records
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch

    dev = torch.device("cuda:0")
    reqs = [(700 + i, 0, 0) for i in range(20)]
    s = DeviceSession(8, 12, 6.6, 0.01, device=dev)
    s.set_games(reqs)
    s.run(hash_eval_torch)
    pos, policy, qp, qn = (t.cpu().numpy() for t in training_tensors(s))
    res = results_from_records([GameMetadata(*r) for r in reqs], s.drain_samples(), s.sample_counts())
    s.close()
    samples = [smp for r in res.results for smp in r.samples]
    samples = samples + [smp.flip_h() for smp in samples]
    assert pos.shape == (len(samples), 2, 6, 7) and policy.shape == (len(samples), 7)
    for i, smp in enumerate(samples):
        a, b, c, d = smp.to_numpy()
        assert np.array_equal(pos[i], a) and np.array_equal(policy[i], b) and qp[i] == c and qn[i] == d


def test_play_games_narrows_the_tail_without_changing_samples():
    """Device mode narrows a session when at most half of its rows still hold a game
    (DeviceSession.narrow_if_worthwhile -> c4_session_compact + a fresh HIP graph).  A run wide
    enough to narrow several times must give the samples of a run too small ever to narrow."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    reqs = [c4a0_amd.GameMetadata(9000 + i, 0, 0) for i in range(1500)]
    st_wide, st_two, st_small = {}, {}, {}
    wide = c4a0_amd.play_games(reqs, 64, 6, 6.6, 0.01, evaluator=net, resident_games=1024, concurrent_sessions=1, stats=st_wide)
    two = c4a0_amd.play_games(reqs, 64, 6, 6.6, 0.01, evaluator=net, resident_games=2048, concurrent_sessions=2, stats=st_two)
    small = c4a0_amd.play_games(reqs, 64, 6, 6.6, 0.01, evaluator=net, resident_games=128, concurrent_sessions=1, stats=st_small)
    a = small.to_arrays()
    for other in (wide, two):
        for x, y in zip(a, other.to_arrays()):
            assert np.array_equal(x, y)
    assert st_wide["sims"] == st_small["sims"] == st_two["sims"]
    assert st_wide["rows_at_end"] < 1024 and st_small["rows_at_end"] == 128


@pytest.mark.parametrize("n_a,n_b", [(900, 300), (300, 900)])
def test_paired_sessions_of_unequal_size_narrow_at_different_checks(n_a, n_b):
    """the two sessions of the paired graph (session._run_pair) reach their narrowing thresholds at
    DIFFERENT checks when they hold different numbers of games; the one that narrows must not be compacted under
    a replay of the shared graph that is still running.  Samples must equal a run that never narrows."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from c4a0_amd.session import DeviceSession, run_sessions

    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    ids_a, ids_b = [(20000 + i, 0, 0) for i in range(n_a)], [(30000 + i, 0, 0) for i in range(n_b)]
    sessions = []
    for ids in (ids_a, ids_b):
        s = DeviceSession(len(ids), 6, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
        s.set_games(ids)
        sessions.append(s)
    rows_seen = []
    orig = DeviceSession.compact

    def spy(self, multiple=256):
        out = orig(self, multiple)
        rows_seen.append((sessions.index(self), self.rows))
        return out

    DeviceSession.compact = spy
    try:
        run_sessions(sessions, net, steps_per_graph=8)
    finally:
        DeviceSession.compact = orig
    got = [s.drain_samples() for s in sessions]
    for s in sessions:
        s.close()
    # both narrowed, and at least once one narrowed alone (a check at which the other kept its width)
    assert {i for i, _ in rows_seen} == {0, 1}, rows_seen
    want = c4a0_amd.play_games([c4a0_amd.GameMetadata(*r) for r in ids_a + ids_b], 64, 6, 6.6, 0.01, evaluator=net,
                               resident_games=128, concurrent_sessions=1)
    want_recs, _ = want.to_records()
    assert np.concatenate(got).tobytes() == want_recs.tobytes()


def test_the_never_reclaimed_arena_still_states_its_limit():
    """Rounds 1-4 refused n_mcts_iterations > 1523 (43 n + 8 blocks per game no longer fit the 16-bit child links).  Round 5
    reclaims the arena above 1 000 iterations (tests/test_gpu_reclaim.py), so the default sizing accepts them; the limit remains,
    with its reason, for a caller that asks for the never-reclaimed arena, whose widest provable search and explicit size still work."""
    from c4a0_amd._lib import C4Error
    from c4a0_amd.session import DeviceSession

    for n in (3000, 2000, 1524):
        with pytest.raises(C4Error, match="n_mcts_iterations > 1523"):
            DeviceSession(2, n, 6.6, 0.01, reclaim=False)
        s = DeviceSession(2, n, 6.6, 0.01)                       # reclaimed: accepted
        assert s.arena()["reclaim_half_blocks"] > n
        s.close()
    s = DeviceSession(2, 1523, 6.6, 0.01, reclaim=False)         # 43 * 1523 + 8 = 65497 blocks
    s.close()
    s = DeviceSession(2, 2000, 6.6, 0.01, blocks_per_slot=65535)  # the caller's own risk, stated in the message
    assert s.arena()["reclaim_half_blocks"] == 0
    s.close()


def test_second_device_after_the_first():
    """sessions, the conv tower's LDS opt-in and the head kernel on device 1 after device 0
    in the same process; the caller's current device is left alone.  Needs two visible devices."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device")
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    reqs = [c4a0_amd.GameMetadata(40 + i, 0, 0) for i in range(12)]
    out = []
    for d in (0, 1):
        dev = torch.device("cuda", d)
        torch.manual_seed(3)
        net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
        before = torch.cuda.current_device()
        out.append(c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=net, device=dev, resident_games=8).to_records())
        assert torch.cuda.current_device() == before
    assert out[0][0].tobytes() == out[1][0].tobytes()


def test_device_callback_wrapper_plays_in_device_mode_with_the_callers_call_unchanged():
    """An unmodified caller passes a callback (training.py:179-189).  A `DeviceCallback` in its place
    answers numpy batches like `forward_numpy` AND is recognised by play_games, which then keeps the
    leaves in HBM: same samples as `evaluator=`, and the callback itself is never invoked."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(21)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 2, 2)), dev, dtype=torch.bfloat16)
    calls = []

    class Spy(c4a0_amd.DeviceCallback):
        def __call__(self, model_id, x):
            calls.append(x.shape[0])
            return super().__call__(model_id, x)

    reqs = [c4a0_amd.GameMetadata(300 + i, 0, 0) for i in range(20)]
    cb = Spy(net, dev)
    via_cb = c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, cb, resident_games=16)          # the reference's six positional arguments
    direct = c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=net, resident_games=16)
    assert calls == [] and via_cb.to_records()[0].tobytes() == direct.to_records()[0].tobytes()
    lp, qp, qn = cb(0, np.zeros((3, 2, 6, 7), dtype=np.float32))                          # and it still is a numpy callback
    assert lp.shape == (3, 7) and qp.shape == (3,) and qn.shape == (3,) and lp.dtype == np.float32 and calls == [3]


def test_leaf_keys_identify_positions():
    """c4_session_leaf_keys: value bits | column heights << 42 per active slot, -1 for idle slots: equal
    keys <=> equal (mask, value)."""
    import ctypes as C
    from c4a0_amd._lib import check
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch

    s = DeviceSession(64, 12, 6.6, 0.01)
    s.set_games([(i, 0, 0) for i in range(40)])     # 24 idle slots
    s.bind()
    s.start()
    keys = torch.zeros(64, dtype=torch.int64, device=s.device)
    for _ in range(30):
        s.evaluate(hash_eval_torch)
        s.step()
    check(s.L.c4_session_leaf_keys(s._h, C.c_void_p(keys.data_ptr())))
    torch.cuda.synchronize()
    mask, value, status = s.leaves()
    k = keys.cpu().numpy()
    s.close()
    for g in range(64):
        if status[g] != 1:
            assert k[g] == -1
            continue
        heights = sum(bin(int(mask[g]) & (0x810204081 << c)).count("1") << (3 * c) for c in range(7))
        assert int(k[g]) == int(value[g]) | (heights << 42)
    act = status == 1
    pos = {(int(m), int(v)) for m, v in zip(mask[act], value[act])}
    assert len(pos) == len(set(k[act].tolist()))


def _unique_reference(mask, value, status, models):
    """What c4_session_unique_leaves must produce, from the host's view of the slots: rows ordered by the
    lowest slot holding each (model, position) pair."""
    rows, inverse, seen = [], np.full(len(mask), 0xFFFFFFFF, dtype=np.uint32), {}
    for g in range(len(mask)):
        if status[g] != 1:
            continue
        pair = (int(models[g]) if models is not None else 0, int(mask[g]), int(value[g]))
        if pair not in seen:
            seen[pair] = len(rows)
            rows.append(g)
        inverse[g] = seen[pair]
    return rows, inverse


@pytest.mark.parametrize("pinned,multi", [(False, False), (True, False), (True, True)])
def test_unique_leaves_and_scatter(pinned, multi):
    """c4_session_unique_leaves + c4_session_scatter_outputs against a host restatement: one row per (model,
    position) pair in lowest-slot order, the planes of c4r.rs:378-392, the slot -> row map, idle slots left out;
    into device memory and into pinned host memory; the answers reach exactly the slots that asked.  Repeated
    over the steps of a running session (the table must come back empty every time)."""
    import ctypes as C
    from c4a0_amd._lib import check
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch

    n = 96
    s = DeviceSession(n, 12, 6.6, 0.01)
    s.set_games([(i, 7 if multi else 0, (1 << 63) + 5 if multi else 0) for i in range(70)])     # 26 idle slots
    models_dev = s.bind_leaf_models() if multi else None
    s.bind()
    s.start()
    dev = s.device
    inverse = torch.zeros(n, dtype=torch.int32, device=dev)
    mk = (lambda *shape, dtype: torch.zeros(*shape, dtype=dtype).pin_memory()) if pinned else (lambda *shape, dtype: torch.zeros(*shape, dtype=dtype, device=dev))
    rows_out, models_out, count = mk(n, 84, dtype=torch.float32), mk(n, dtype=torch.int64), mk(1, dtype=torch.int32)
    answers = mk(n, 9, dtype=torch.float32)
    for step in range(25):
        s.evaluate(hash_eval_torch)
        s.step()
        if step % 4:
            continue
        check(s.L.c4_session_unique_leaves(s._h, C.c_void_p(inverse.data_ptr()), C.c_void_p(rows_out.data_ptr()),
                                           C.c_void_p(models_out.data_ptr()) if multi else None, C.c_void_p(count.data_ptr())))
        torch.cuda.synchronize()
        mask, value, status = s.leaves()
        models = models_dev.cpu().numpy() if multi else None
        want_rows, want_inverse = _unique_reference(mask, value, status, models)
        n_u = int(count.cpu()[0])
        assert n_u == len(want_rows) and 0 < n_u < int((status == 1).sum())         # hash_eval games do collide early on
        assert np.array_equal(inverse.cpu().numpy().view(np.uint32), want_inverse)
        got = rows_out.cpu().numpy()[:n_u]
        for r, g in enumerate(want_rows):
            m, v = int(mask[g]), int(value[g])
            mine = [(v >> i) & 1 for i in range(42)]
            theirs = [((m ^ v) >> i) & 1 for i in range(42)]
            assert got[r].tolist() == [float(b) for b in mine + theirs]
        if multi:
            assert np.array_equal(models_out.cpu().numpy()[:n_u], models[want_rows])
        # answers: row r carries r + e / 16 in element e
        a = (torch.arange(n_u, dtype=torch.float32)[:, None] + torch.arange(9, dtype=torch.float32)[None, :] / 16)
        answers[:n_u] = a if pinned else a.to(dev)
        before_lp, before_q = s.logprobs.clone(), s.q.clone()
        check(s.L.c4_session_scatter_outputs(s._h, C.c_void_p(inverse.data_ptr()), C.c_void_p(answers.data_ptr()), n_u))
        torch.cuda.synchronize()
        lp, q = s.logprobs.cpu().numpy(), s.q.cpu().numpy()
        for g in range(n):
            if want_inverse[g] == 0xFFFFFFFF:
                assert np.array_equal(lp[g], before_lp[g].cpu().numpy()) and np.array_equal(q[g], before_q[g].cpu().numpy())
            else:
                r = float(want_inverse[g])
                assert lp[g].tolist() == [r + e / 16 for e in range(7)] and q[g].tolist() == [r + 7 / 16, r + 8 / 16]
    # pageable host memory is refused, by name
    bad = np.zeros((n, 84), dtype=np.float32)
    rc = s.L.c4_session_unique_leaves(s._h, C.c_void_p(inverse.data_ptr()), C.c_void_p(bad.ctypes.data), None, C.c_void_p(count.data_ptr()))
    assert rc != 0 and b"rows_out" in s.L.c4_last_error_string()
    s.close()


def test_arena_is_kept_for_the_next_session_and_can_be_trimmed():
    """c4_session_destroy keeps one tree arena for the next session that fits it (the driver scrubs freed memory
    before reuse: 0.6 s for the reference job's 13 GB); a reused arena holds the previous games' trees and nothing
    of them may show: same samples as the oracle, three sessions in a row of sizes that do and do not fit;
    trim_cached_memory() returns the memory (the device's free bytes go up by about the arena's size)."""
    import c4a0_amd
    from oracle import c4oracle as O
    from tests.helpers import hash_eval_np, oracle_samples_by_game

    c4a0_amd.trim_cached_memory()
    free0, _ = torch.cuda.mem_get_info()
    for n_games, resident, n_iter in ((40, 32, 60), (24, 24, 50), (70, 64, 9), (40, 32, 60)):
        reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n_games)]
        got = c4a0_amd.play_games(reqs, 64, n_iter, 6.6, 0.01, hash_eval_np, resident_games=resident)
        want, _ = O.self_play([(i, 0, 0) for i in range(n_games)], 64, n_iter, 6.6, 0.01, "hash")
        assert _as_oracle_dict(got) == oracle_samples_by_game(want)
    free1, _ = torch.cuda.mem_get_info()
    c4a0_amd.trim_cached_memory()
    free2, _ = torch.cuda.mem_get_info()
    arena = 32 * (43 * 60 + 8) * 128                      # the biggest of the four
    assert free2 - free1 >= arena // 2 and free0 - free1 >= arena // 2, (free0, free1, free2, arena)
    c4a0_amd.trim_cached_memory()                          # nothing kept: still fine


def test_an_evaluator_that_fails_mid_job_poisons_nothing():
    """The caller's evaluator raises in the middle of a job -- a numpy callback, a device callable launched eagerly, a graph-safe one
    (which may be inside a HIP-graph capture at that moment): the exception reaches the caller (the reference panics,
    pybridge.rs:182-183), the sessions are closed, and the next job returns the bytes it always returns."""
    import c4a0_amd
    from c4a0_amd._lib import lib
    from tests.helpers import hash_eval_np, hash_eval_torch

    reqs = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(300)]
    want = c4a0_amd.play_games(reqs, 64, 12, 6.6, 0.01, hash_eval_np).to_records()[0].tobytes()

    def free_bytes():
        torch.cuda.synchronize()
        lib().c4_trim_cached_memory()
        torch.cuda.empty_cache()
        return torch.cuda.mem_get_info()[0]

    calls = {"np": 0, "graph": 0, "eager": 0}

    def np_cb(model_id, pos):
        calls["np"] += 1
        if calls["np"] == 7:
            raise RuntimeError("boom (numpy callback)")
        return hash_eval_np(model_id, pos)

    def graph_safe(planes, out_logprobs=None, out_q=None):
        calls["graph"] += 1
        if calls["graph"] == 5:
            raise RuntimeError("boom (graph-safe device callable)")
        lp, q = hash_eval_torch(planes)
        out_logprobs.copy_(lp)
        out_q.copy_(q)

    graph_safe.graph_safe = True

    def eager(planes):
        calls["eager"] += 1
        if calls["eager"] == 40:
            raise RuntimeError("boom (eager device callable)")
        return hash_eval_torch(planes)

    free0 = None
    for kw in (dict(py_eval_pos_cb=np_cb), dict(evaluator=graph_safe), dict(evaluator=eager)):
        with pytest.raises(RuntimeError, match="boom"):
            c4a0_amd.play_games(reqs, 64, 12, 6.6, 0.01, **kw)
        again = c4a0_amd.play_games(reqs, 64, 12, 6.6, 0.01, evaluator=hash_eval_torch).to_records()[0].tobytes()
        assert again == want, kw
        if free0 is None:
            free0 = free_bytes()                       # (after the first failure: one-time pools of torch's graph allocator are in)
    assert free0 - free_bytes() < 128 << 20
