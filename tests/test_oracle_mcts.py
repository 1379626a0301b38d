"""Oracle vs the reference's MCTS / softmax / temperature / self-play known-answer tests
(rust/src/mcts.rs:463-686, rust/proptest-regressions/mcts.txt, rust/src/self_play.rs:389-459)."""
import math
import random

import numpy as np
import pytest

from oracle import c4oracle as O

F = np.float32
CONST_COL_WEIGHT = F(1.0) / F(7.0)       # mcts.rs:465
EPS = 1e-8                               # Node::EPS mcts.rs:343
C_EXPL, C_PLY = 4.0, 0.01                # mcts.rs:466-467


def run(pos, n):
    p, qp, qn, _g = O.run_mcts(pos, n, C_EXPL, C_PLY)
    return p, qp, qn


def assert_policy_sum_1(p):              # mcts.rs:688-693
    assert abs(float(np.sum(p.astype(np.float32), dtype=np.float32)) - 1.0) <= 1e-5


def test_mcts_prefers_center_column():   # mcts.rs:488-492
    p, _, _ = run(O.Pos(0, 0), 1000)
    assert_policy_sum_1(p)
    assert p[3] > CONST_COL_WEIGHT
    # SURVEY 8c derived golden value (independent restatement): [0.14214215 x2, 0.14314315 x5]
    assert np.allclose(p, [0.14214215] * 2 + [0.14314315] * 5, atol=1e-7)


def test_mcts_depth_one():               # mcts.rs:495-499
    p, _, _ = run(O.Pos(0, 0), 1 + 7 + 7)
    assert np.all(np.abs(p - CONST_COL_WEIGHT) < EPS)


def test_mcts_depth_two():               # mcts.rs:502-508
    p, _, _ = run(O.Pos(0, 0), 1 + 7 + 49 + 49)
    assert np.all(np.abs(p - CONST_COL_WEIGHT) < EPS)


def test_mcts_depth_uneven():            # mcts.rs:511-514
    p, _, _, g = O.run_mcts(O.Pos(0, 0), 47, C_EXPL, C_PLY)
    assert np.any(np.abs(p - CONST_COL_WEIGHT) > EPS)
    # child visits [6,6,6,7,7,7,7] -- shows the LAST-maximum tie-break (mcts.rs:165-173)
    assert np.array_equal(np.round(p * 46).astype(int), [6, 6, 6, 7, 7, 7, 7])


def test_winning_position():             # mcts.rs:519-538
    pos = O.from_rows(["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫🔵🔵🔵⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫"])
    p, qp, qn = run(pos, 10_000)
    assert math.isclose(float(p.sum(dtype=np.float32)), 1.0, rel_tol=1e-6)
    assert p[0] + p[4] > 0.99 and qp > 0.92 and qn > 0.99
    # SURVEY 8c derived values
    assert np.allclose(p, [0.49924994, 3.0003e-4, 3.0003e-4, 3.0003e-4, 0.49924994, 3.0003e-4, 3.0003e-4], atol=1e-7)
    assert abs(qp - 0.92845) < 1e-5 and abs(qn - 0.99830) < 1e-5


def test_winning_position2():            # mcts.rs:541-560
    pos = O.from_rows(["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫⚫🔵🔵⚫⚫⚫", "⚫⚫🔴🔴⚫⚫⚫"])
    p, qp, qn = run(pos, 10_000)
    assert p[1] + p[4] > 0.98 and qp > 0.90 and qn > 0.98 and qn > qp


def test_winning_position3():            # mcts.rs:563-581
    pos = O.from_rows(["⚫⚫⚫⚫⚫⚫⚫"] * 3 + ["⚫🔴🔵🔵⚫⚫⚫", "⚫🔵🔴🔴🔴⚫⚫", "⚫🔵🔵🔴🔵🔴⚫"])
    p, qp, qn = run(pos, 10_000)
    assert p[5] > 0.99 and qp > 0.86 and qn > 0.99 and qn > qp


def test_losing_position():              # mcts.rs:585-607
    pos = O.from_rows(["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫🔴🔴⚫⚫⚫⚫", "⚫🔵🔵🔵⚫⚫⚫"])
    p, qp, qn = run(pos, 300_000)
    assert_policy_sum_1(p)
    assert np.all(np.abs(p - CONST_COL_WEIGHT) <= 0.01)
    assert qp < -0.93 and qn < -0.99 and qn < qp


def test_prefer_shorter_wins():          # mcts.rs:611-632
    pos = O.from_rows(["⚫⚫⚫🔵⚫⚫⚫", "⚫🔵🔵🔵⚫⚫⚫", "⚫🔴🔵🔵⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫", "⚫🔵🔴🔵⚫⚫⚫"])
    p, qp, qn = run(pos, 10_000)
    assert p[4] > 0.99 and qp > 0.82 and qn > 0.99 and qn > qp


# ---- softmax / temperature properties, mcts.rs:635-686 -------------------------------------
def policy_strategy(rng):                # mcts.rs:635-645
    while True:
        logits = [(-math.inf if rng.random() < 0.5 else float(F(rng.uniform(0.0, 10.0)))) for _ in range(7)]
        if not all(l == -math.inf for l in logits):
            return O.softmax7(logits)


REGRESSION_POLICIES = [                  # rust/proptest-regressions/mcts.txt:7-12
    ("policy", [0.0] * 7),
    ("policy_log", [0.0, 0.0, -6.872888e19, 0.0, 0.0, 0.0, 0.0]),
    ("policy", [0.9286058, 0.0, 0.0033046294, 0.06687763, 0.0, 0.0, 0.001211846]),
    ("policy", [0.4780801, 2.5148089e-5, 2.5148089e-5, 0.52179414, 2.5148089e-5, 2.5148089e-5, 2.5148089e-5]),
    ("policy", [0.0, 0.933416, 0.00035163847, 0.0009350313, 0.0, 0.06520966, 8.7709726e-5]),
    ("policy", [0.0, 0.106206864, 0.0, 0.0, 0.148644, 0.7410872, 0.004062006]),
]


def _check_props(policy):
    policy = np.asarray(policy, dtype=np.float32)
    t1 = O.apply_temperature(policy, 1.0)                   # temperature_1, mcts.rs:655-658
    assert np.all(np.abs(t1 - policy) < 1e-5)
    if abs(float(policy.sum(dtype=np.float32)) - 1.0) <= 1e-5:
        t2 = O.apply_temperature(policy, 2.0)               # temperature_2, mcts.rs:662-670
        assert_policy_sum_1(t2)
        if sum(1 for p in policy if p != CONST_COL_WEIGHT and p > 0.0) >= 2:
            assert np.any(np.abs(t2 - policy) > EPS)
        assert np.all(t2[policy == 0.0] == 0.0)
    t0 = O.apply_temperature(policy, 0.0)                   # temperature_0, mcts.rs:674-685
    mx = t0.max()
    cnt = int((t0 == mx).sum())
    assert_policy_sum_1(t0)
    assert all(p == F(1.0) / F(cnt) for p in t0 if p == mx)


def test_softmax_and_temperature_properties():
    rng = random.Random(1337)
    for _ in range(2000):
        p = policy_strategy(rng)
        assert_policy_sum_1(p)                              # softmax_sum_1, mcts.rs:649-651
        _check_props(p)


def test_softmax_temperature_regressions():
    for kind, vec in REGRESSION_POLICIES:
        if kind == "policy_log":
            p = O.softmax7(vec)
            assert_policy_sum_1(p)
            assert p[2] == 0.0
            _check_props(p)
        elif sum(vec) > 0:
            _check_props(vec)
        else:
            # all-zero policy: all entries equal => apply_temperature is the identity (mcts.rs:440)
            assert np.array_equal(O.apply_temperature(vec, 2.0), np.zeros(7, np.float32))


def test_softmax_degenerate_is_an_error():                  # mcts.rs:421-425 panic
    with pytest.raises(ValueError):
        O.softmax7([-math.inf] * 7)
    with pytest.raises(ValueError):
        O.softmax7([0, 0, math.inf, 0, 0, 0, 0])


def test_softmax_matches_numpy_f32_with_host_libm():
    """softmax restated with the host libm's expf must agree bit for bit."""
    rng = np.random.default_rng(5)
    L = O.lib()
    import ctypes as C
    for _ in range(500):
        x = rng.normal(0, 3, 7).astype(np.float32)
        mx = x.max()
        d = (x - mx).astype(np.float32)
        e = np.empty(7, np.float32)
        L.c4o_host_expf(d.ctypes.data_as(C.POINTER(C.c_float)), e.ctypes.data_as(C.POINTER(C.c_float)), 7)
        s = F(0.0)
        for v in e:
            s = F(s + v)
        assert np.array_equal(O.softmax7(x), (e / s).astype(np.float32))


# ---- self_play.rs:389-459 ------------------------------------------------------------------
def test_self_play_structure():                             # self_play.rs:405-458 (UniformEvalPos)
    res, st = O.self_play([(0, 0, 0)], 10, 50, 1.0, 0.01, "uniform")
    assert st["n_games"] == 1
    for gid, samples in res.items():
        assert len(samples) >= 7
        assert sum(1 for s in samples if (s.mask, s.value) == (0, 0)) == 1
        term = [s for s in samples if O.terminal_state(O.Pos(s.mask, s.value)) != 0]
        assert len(term) == 1
        assert term[0].q_no_penalty in (-1.0, 0.0, 1.0)
        assert term[0] is samples[-1]
        assert term[0].policy == tuple(float(CONST_COL_WEIGHT) for _ in range(7))   # mcts.rs:45,300-305


def test_self_play_batch_cap_and_dedup():                   # self_play.rs:216-220, 394
    seen = []

    def ev(model_id, x):
        seen.append(x.shape[0])
        assert len({x[i].tobytes() for i in range(x.shape[0])}) == x.shape[0]  # unique positions per call
        b = x.shape[0]
        return np.zeros((b, 7), np.float32), np.zeros(b, np.float32), np.zeros(b, np.float32)

    res, st = O.self_play([(i, 0, 0) for i in range(12)], 5, 4, 1.4, 0.01, ev)
    assert max(seen) <= 5 and len(res) == 12
    assert seen[0] == 1  # all 12 games start on the empty board: one unique position


def test_self_play_q_signs_and_threads():                   # mcts.rs:279-298
    reqs = [(i, 0, 0) for i in range(16)]
    res1, st1 = O.self_play(reqs, 64, 20, 6.6, 0.01, "hash", n_threads=1)
    res4, st4 = O.self_play(reqs, 64, 20, 6.6, 0.01, "hash", n_threads=4)
    assert res1 == res4 and st1 == st4
    for gid, s in res1.items():
        m = len(s) - 1
        qp, qn = s[-1].q_penalty, s[-1].q_no_penalty
        for i in range(m):
            sign = 1.0 if (m - i) % 2 == 0 else -1.0
            assert s[i].q_penalty == sign * qp and s[i].q_no_penalty == sign * qn
        # consecutive positions are one move apart
        for a, b in zip(s[:-1], s[1:]):
            assert bin(b.mask).count("1") == bin(a.mask).count("1") + 1
            assert any((nx := O.make_move(O.Pos(a.mask, a.value), c)) is not None and nx.key() == (b.mask, b.value) for c in range(7))


def test_multi_model_majority():                            # self_play.rs:203-215, mcts.rs:70-76
    calls = []

    def ev(model_id, x):
        calls.append((model_id, x.shape[0]))
        b = x.shape[0]
        return np.full((b, 7), float(model_id), np.float32), np.zeros(b, np.float32), np.zeros(b, np.float32)

    res, _ = O.self_play([(0, 1, 2), (1, 2, 1), (2, 1, 2)], 8, 3, 1.4, 0.01, ev)
    assert {m for m, _ in calls} == {1, 2} and len(res) == 3


# ---- Dirichlet root noise: build extension (BASELINE.json names it; the reference has none) ----
def test_dirichlet_extension_basic_properties():
    for alpha in (0.03, 0.3, 1.0, 1.4, 10.0):
        for g in range(50):
            legal = 0x7F if g % 3 else 0b1011101
            e = O.dirichlet(g, g % 40, legal, alpha)
            assert abs(float(e.sum(dtype=np.float32)) - 1.0) < 1e-5 and e.min() >= 0.0
            assert all(e[c] == 0.0 for c in range(7) if not (legal >> c) & 1)
    # deterministic in (game_id, n_moves); different across moves
    assert np.array_equal(O.dirichlet(9, 4, 0x7F, 0.3), O.dirichlet(9, 4, 0x7F, 0.3))
    assert not np.array_equal(O.dirichlet(9, 4, 0x7F, 0.3), O.dirichlet(9, 5, 0x7F, 0.3))
    # component means of Dir(alpha) over 7 columns are 1/7
    m = np.mean([O.dirichlet(g + 1, 0, 0x7F, 0.3) for g in range(3000)], axis=0)
    assert np.all(np.abs(m - 1 / 7) < 0.02)


def test_dirichlet_off_by_default_and_changes_games_when_on():
    reqs = [(i, 0, 0) for i in range(6)]
    a, _ = O.self_play(reqs, 64, 15, 6.6, 0.01, "hash")
    b, _ = O.self_play(reqs, 64, 15, 6.6, 0.01, "hash", dirichlet=(0.3, 0.0))
    c, _ = O.self_play(reqs, 64, 15, 6.6, 0.01, "hash", dirichlet=(0.3, 0.25))
    d, _ = O.self_play(reqs, 64, 15, 6.6, 0.01, "hash")
    assert a == b == d and a != c
    for s in c.values():   # noisy games are still well-formed
        assert O.terminal_state(O.Pos(s[-1].mask, s[-1].value)) != 0


def test_async_topology_gives_the_samples_of_the_lockstep_restatement():
    """c4o_self_play_async = the reference's thread topology (self_play.rs:60-106: one NN thread,
    ncpu-1 MctsThreads, two queues).  Batching depends on thread timing, a game's samples do not
    (each trajectory depends on the answers for its own leaves only): same samples and tree
    counters as the lock-step restatement, for built-in and Python evaluators, several models,
    batch caps smaller than the job, and more threads than games."""
    from tests.helpers import hash_eval_np

    def flat(res):
        return {g: [(s.mask, s.value, tuple(s.policy), s.q_penalty, s.q_no_penalty) for s in ss] for g, ss in res.items()}

    reqs = [(3 * i + 1, i % 3, (i + 1) % 3) for i in range(90)]
    want, wst = O.self_play(reqs, 7, 15, 6.6, 0.01, "hash")
    for threads in (2, 5):
        got, gst = O.self_play(reqs, 7, 15, 6.6, 0.01, "hash", n_threads=threads, topology="async")
        assert flat(got) == flat(want)
        for k in ("sims", "backup_nodes", "expansions", "moves", "n_samples"):
            assert gst[k] == wst[k], k
    got, _ = O.self_play(reqs[:20], 64, 9, 6.6, 0.01, hash_eval_np, n_threads=3, topology="async")
    want2, _ = O.self_play(reqs[:20], 64, 9, 6.6, 0.01, "hash")
    assert flat(got) == flat(want2)
    got, _ = O.self_play(reqs[:2], 64, 9, 6.6, 0.01, "hash", n_threads=8, topology="async")
    assert flat(got) == {g: v for g, v in flat(want2).items() if g in (1, 4)}
    assert O.self_play([], 64, 9, 6.6, 0.01, "hash", n_threads=4, topology="async")[0] == {}

    # regression (round 2): oversubscribed threads and tiny rings -- a producer that laps a preempted
    # consumer must wait for its cell, not drop the game (the job used to hang now and then)
    for _ in range(150):
        got, gst = O.self_play(reqs[:5], 64, 9, 6.6, 0.01, "uniform", n_threads=16, topology="async")
        assert gst["n_games"] == 5 and len(got) == 5

    def failing(_m, _x):
        raise RuntimeError("evaluator failure")
    with pytest.raises(RuntimeError):
        O.self_play(reqs[:8], 64, 9, 6.6, 0.01, failing, n_threads=3, topology="async")


def test_table_evaluator_replays_a_run_from_its_logged_answers():
    """c4o_eval_table (tier T3 at full size, tests/test_gpu_baseline_configs.py): an evaluator that answers a position with what another
    evaluator said for it.  A run through a callback that logs (position -> answer) and the replay from the sorted table give the same
    samples on 1 and 4 threads; a table that lacks non-terminal positions fails the replay instead of inventing an answer."""
    from tests.helpers import hash_eval_np, oracle_samples_by_game, planes_to_pos_np

    reqs = [(g, 0, 0) for g in range(24)]
    seen = {}

    def cb(model_id, x):
        lp, qp, qn = hash_eval_np(model_id, x)
        m, v = planes_to_pos_np(x)
        for i in range(len(m)):
            seen[(int(m[i]), int(v[i]))] = np.concatenate([lp[i], [qp[i]], [qn[i]]]).astype(np.float32)
        return lp, qp, qn

    want, _ = O.self_play(reqs, 64, 25, 6.6, 0.01, cb)
    keys = sorted(k for k in seen if O.terminal_state(O.Pos(*k)) == 0)          # terminal leaves: asked, never used (mcts.rs:92-98)
    assert len(keys) < len(seen)
    mask = np.array([k[0] for k in keys], dtype=np.uint64)
    value = np.array([k[1] for k in keys], dtype=np.uint64)
    out = np.stack([seen[k] for k in keys])
    for threads, topology in ((1, "lockstep"), (4, "async")):
        got, _ = O.self_play(reqs, 64, 25, 6.6, 0.01, ("table", mask, value, out), n_threads=threads, topology=topology)
        assert oracle_samples_by_game(got) == oracle_samples_by_game(want)
    with pytest.raises(RuntimeError):
        O.self_play(reqs, 64, 25, 6.6, 0.01, ("table", mask[:-7], value[:-7], out[:-7]))
