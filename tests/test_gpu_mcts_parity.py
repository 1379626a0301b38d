"""GPU parity of the tree path: the fused HIP step kernel vs the CPU oracle, bit for bit,
through the C ABI.  Tiers follow SURVEY 8c: T2 = the reference's own MCTS known-answer tests
run ON THE DEVICE (constant evaluator), T1 = whole self-play games under the integer-hash
evaluator, compared per game_id sample by sample."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

F = np.float32
UNIFORM = float(F(1.0) / F(7.0))


@pytest.fixture(scope="module")
def env():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from c4a0_amd.session import DeviceSession
    from oracle import c4oracle as O

    return DeviceSession, O, torch.device("cuda:0")


def device_run_mcts(DeviceSession, pos, n_iter, c_expl=4.0, c_ply=0.01, blocks=None):
    """mcts.rs:469-485 `run_mcts` on the GPU: one game, constant evaluator, no moves."""
    from tests.helpers import uniform_eval_torch

    s = DeviceSession(1, 1 << 30, c_expl, c_ply, blocks_per_slot=blocks or (n_iter + 8), no_moves=True)
    s.set_games([(0, 0, 0)], [pos])
    s.bind()
    s.start()
    s.evaluate(uniform_eval_torch)  # constant: evaluate once, the bound tensors never change
    for _ in range(n_iter):
        s.step()
    pol, qp, qn, n, root = s.root_stats(0)
    c = s.counters()
    s.close()
    assert c["error"] == 0 and n == n_iter and root == pos
    return pol, qp, qn, c


def _same(a, b):
    return np.array_equal(np.asarray(a, dtype=np.float32).view(np.uint32), np.asarray(b, dtype=np.float32).view(np.uint32))


@pytest.mark.parametrize("n_iter", [1, 2, 15, 47, 106, 1000])
def test_empty_board_kats_match_oracle(env, n_iter):
    # mcts.rs:488-514 (prefers_center 1000, depth_one 15, depth_two 106, depth_uneven 47)
    DeviceSession, O, dev = env
    pol, qp, qn, c = device_run_mcts(DeviceSession, (0, 0), n_iter)
    opol, oqp, oqn, g = O.run_mcts(O.Pos(0, 0), n_iter, 4.0, 0.01)
    assert _same(pol, opol) and _same([qp, qn], [oqp, oqn])
    oc = g.counters()
    assert (c["sims"], c["select_levels"], c["backup_nodes"], c["expansions"]) == \
           (oc["sims"], oc["select_levels"], oc["backup_nodes"], oc["expansions"])
    if n_iter in (15, 106):
        assert np.all(np.abs(pol - F(UNIFORM)) < 1e-8)
    if n_iter == 47:
        assert np.array_equal(np.round(pol * 46).astype(int), [6, 6, 6, 7, 7, 7, 7])  # last-max tie-break
    if n_iter == 1000:
        assert pol[3] > F(UNIFORM)


TACTICAL = [  # (rows, n_iter, check) -- mcts.rs:519-632
    (["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫🔵🔵🔵⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫"], 10_000,
     lambda p, qp, qn: p[0] + p[4] > 0.99 and qp > 0.92 and qn > 0.99),
    (["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫⚫🔵🔵⚫⚫⚫", "⚫⚫🔴🔴⚫⚫⚫"], 10_000,
     lambda p, qp, qn: p[1] + p[4] > 0.98 and qp > 0.90 and qn > 0.98 and qn > qp),
    (["⚫⚫⚫⚫⚫⚫⚫"] * 3 + ["⚫🔴🔵🔵⚫⚫⚫", "⚫🔵🔴🔴🔴⚫⚫", "⚫🔵🔵🔴🔵🔴⚫"], 10_000,
     lambda p, qp, qn: p[5] > 0.99 and qp > 0.86 and qn > 0.99 and qn > qp),
    (["⚫⚫⚫🔵⚫⚫⚫", "⚫🔵🔵🔵⚫⚫⚫", "⚫🔴🔵🔵⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫", "⚫🔴🔴🔴⚫⚫⚫", "⚫🔵🔴🔵⚫⚫⚫"], 10_000,
     lambda p, qp, qn: p[4] > 0.99 and qp > 0.82 and qn > 0.99 and qn > qp),
    (["⚫⚫⚫⚫⚫⚫⚫"] * 4 + ["⚫🔴🔴⚫⚫⚫⚫", "⚫🔵🔵🔵⚫⚫⚫"], 30_000,   # losing_position, shortened from 300k
     lambda p, qp, qn: qp < -0.9 and qn < -0.95 and qn < qp),
]


@pytest.mark.parametrize("case", range(len(TACTICAL)))
def test_tactical_kats_match_oracle(env, case):
    DeviceSession, O, dev = env
    rows, n_iter, check = TACTICAL[case]
    pos = O.from_rows(rows)
    pol, qp, qn, c = device_run_mcts(DeviceSession, pos.key(), n_iter)
    opol, oqp, oqn, g = O.run_mcts(pos, n_iter, 4.0, 0.01)
    assert check(pol, qp, qn)
    assert _same(pol, opol) and _same([qp, qn], [oqp, oqn])


def _play(DeviceSession, reqs, n_slots, n_iter, c_expl, c_ply, evaluator, planes_dtype=torch.float32):
    s = DeviceSession(n_slots, n_iter, c_expl, c_ply, planes_dtype=planes_dtype)
    s.set_games(reqs)
    steps = s.run(evaluator, max_steps=2_000_000)
    recs = s.drain_samples()
    c = s.counters()
    counts = s.sample_counts()
    s.close()
    return recs, c, counts, steps


@pytest.mark.parametrize("n_games,n_slots,n_iter,c_expl", [
    (32, 32, 10, 6.6),     # BASELINE config 1 shape (32 games, n_mcts = 10)
    (96, 24, 25, 6.6),     # more games than slots: finished games are replaced mid-run
    (40, 64, 5, 1.4),      # fewer games than slots: idle slots
    (16, 16, 100, 6.6),    # n_mcts = 100 (BASELINE metric setting), few games
    (8, 8, 2, 4.0),        # smallest gate with a visited child: a move every other simulation
])
def test_self_play_hash_evaluator_bit_identical(env, n_games, n_slots, n_iter, c_expl):
    """T1: every sample of every game identical to the oracle (per game_id, order-free)."""
    DeviceSession, O, dev = env
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game

    # ids include 0 (seed 0 on every move) and colliding seeds 43*42 == 42*43 (mcts.rs:215)
    ids = [0, 42, 43, 1 << 40, (1 << 64) - 1][: min(5, n_games)] + list(range(1000, 1000 + n_games))
    reqs = [(gid, 0, 0) for gid in ids[:n_games]]
    recs, c, counts, steps = _play(DeviceSession, reqs, n_slots, n_iter, c_expl, 0.01, hash_eval_torch)
    ores, ost = O.self_play(reqs, 1 << 20, n_iter, c_expl, 0.01, "hash")
    got, want = samples_by_game(recs), oracle_samples_by_game(ores)
    assert set(got) == set(want)
    for gid in want:
        assert got[gid] == want[gid], f"game {gid} differs"
    assert c["games_done"] == n_games and c["samples"] == ost["n_samples"] == len(recs)
    assert np.array_equal(counts, [len(ores[g]) for g, _, _ in reqs])
    # device counters are the exact roofline numerators: they must equal the oracle's
    assert c["sims"] + c["ref_skipped_sims"] == ost["sims"]
    assert c["ref_skipped_sims"] == ost["sims_terminal_root"]
    # the device does not run the select whose leaf a move discards in the same job
    assert c["select_levels"] == ost["select_levels"] - ost["select_levels_discarded"]
    assert c["backup_nodes"] == ost["backup_nodes"]
    assert c["expansions"] == ost["expansions"]
    assert c["moves"] == ost["moves"]


def test_self_play_uniform_evaluator_structure_and_parity(env):
    """T2 + self_play.rs:405-458 structural invariants, on the device, bf16 planes."""
    DeviceSession, O, dev = env
    from tests.helpers import oracle_samples_by_game, samples_by_game, uniform_eval_torch

    reqs = [(i, 0, 0) for i in range(48)]
    recs, c, counts, _ = _play(DeviceSession, reqs, 16, 50, 1.0, 0.01, uniform_eval_torch, planes_dtype=torch.bfloat16)
    ores, _ = O.self_play(reqs, 10, 50, 1.0, 0.01, "uniform")
    got, want = samples_by_game(recs), oracle_samples_by_game(ores)
    assert got == want
    for gid, ss in got.items():
        assert len(ss) >= 7
        assert sum(1 for s in ss if (s[0], s[1]) == (0, 0)) == 1
        term = [s for s in ss if O.terminal_state(O.Pos(s[0], s[1])) != 0]
        assert len(term) == 1 and term[0] is ss[-1]
        assert np.frombuffer(term[0][4], dtype=np.float32)[0] in (-1.0, 0.0, 1.0)


def test_errors_are_raised_not_hidden(env):
    """Reference panics become status codes: NaN in UCT (utils.rs:12), all -inf policy
    (mcts.rs:421-425), arena overflow (build-specific)."""
    DeviceSession, O, dev = env
    from c4a0_amd._lib import C4Error

    def nan_eval(planes):
        g = planes.shape[0]
        return torch.full((g, 7), float("nan"), device=planes.device), torch.zeros((g, 2), device=planes.device)

    def neginf_eval(planes):
        g = planes.shape[0]
        return torch.full((g, 7), float("-inf"), device=planes.device), torch.zeros((g, 2), device=planes.device)

    def nan_q_eval(planes):
        g = planes.shape[0]
        return torch.zeros((g, 7), device=planes.device), torch.full((g, 2), float("nan"), device=planes.device)

    from tests.helpers import uniform_eval_torch
    for ev, code, kw in ((nan_eval, 4, {}), (neginf_eval, 4, {}), (nan_q_eval, 3, {}),
                         (uniform_eval_torch, 5, {"blocks_per_slot": 4})):
        s = DeviceSession(4, 10, 6.6, 0.01, **kw)
        s.set_games([(i, 0, 0) for i in range(4)])
        with pytest.raises(C4Error) as ei:
            s.run(ev, max_steps=5000, poll_every=4)
        assert ei.value.status == code
        s.close()


def test_gate_of_one_iteration_matches_reference_behaviour(env):
    """n_mcts_iterations = 1: the root's children never have visits, root_policy falls back to
    UNIFORM over all 7 columns (mcts.rs:404-406) and a full column can be sampled, where the
    reference panics (mcts.rs:196-200).  Device and oracle must agree game by game: same
    samples, or the same failure."""
    DeviceSession, O, dev = env
    from c4a0_amd._lib import C4Error, ERR_ILLEGAL_MOVE
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game

    n_fail = 0
    for gid in range(40):
        reqs = [(gid, 0, 0)]
        try:
            ores, _ = O.self_play(reqs, 64, 1, 4.0, 0.01, "hash")
            want = oracle_samples_by_game(ores)
        except RuntimeError:
            want = None
        s = DeviceSession(1, 1, 4.0, 0.01)
        s.set_games(reqs)
        try:
            s.run(hash_eval_torch, max_steps=500, poll_every=1)
            got = samples_by_game(s.drain_samples())
        except C4Error as e:
            assert e.status == ERR_ILLEGAL_MOVE
            got = None
        s.close()
        assert got == want, gid
        n_fail += want is None
    assert 0 <= n_fail < 40


def test_pack_samples_on_device_equals_host_drain(env):
    """K6 (c4_session_pack_samples): finished games' records packed on the device, in request
    order, equal the host-side drain -- including while some games are still unfinished."""
    DeviceSession, O, dev = env
    from c4a0_amd.session import SAMPLE_DTYPE
    from tests.helpers import hash_eval_torch

    s = DeviceSession(8, 6, 6.6, 0.01)
    s.set_games([(i, 0, 0) for i in range(30)])
    s.bind()
    s.start()
    for _ in range(90):                       # stop mid-run: some games finished, some not
        s.evaluate(hash_eval_torch)
        s.step()
    for rnd in range(2):
        host = s.drain_samples()
        packed = s.pack_samples_device().cpu().numpy().reshape(-1).view(SAMPLE_DTYPE)
        assert 0 < len(host) and host.tobytes() == packed.tobytes()
        counts = s.sample_counts()
        assert counts.sum() == len(host) and ((counts == 0).any() if rnd == 0 else (counts > 0).all())
        for _ in range(4000):
            s.evaluate(hash_eval_torch)
            s.step()
    assert (s.sample_counts() > 0).all()
    s.close()


def test_dirichlet_noise_extension_matches_its_specification(env):
    """Dirichlet root noise is a build extension (BASELINE.json names it; the reference has none):
    the specification is the oracle's c4o_dirichlet.  The device sampler must equal it bit for bit,
    whole noisy self-play games must equal the oracle's, and with epsilon = 0 nothing changes."""
    import ctypes as C

    DeviceSession, O, dev = env
    from c4a0_amd import _lib
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game

    L = _lib.lib()
    rng = np.random.default_rng(2)
    n = 3000
    gid = rng.integers(0, 1 << 50, n).astype(np.uint64)
    gid[:5] = [0, 1, 42, 43, (1 << 64) - 1]
    nm = rng.integers(0, 42, n).astype(np.uint32)
    legal = rng.integers(1, 128, n).astype(np.uint32)
    for alpha in (1.0, 0.3, 0.03, 1.4, 10.0):
        tg = torch.from_numpy(gid.view(np.int64)).to(dev)
        tn = torch.from_numpy(nm.view(np.int32)).to(dev)
        tl = torch.from_numpy(legal.view(np.int32)).to(dev)
        out = torch.empty((n, 7), dtype=torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(L.c4_dirichlet(p(tg), p(tn), p(tl), alpha, n, p(out), None))
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        for i in range(0, n, 3):
            want = O.dirichlet(int(gid[i]), int(nm[i]), int(legal[i]), alpha)
            assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (alpha, i)
        assert np.all(np.abs(got.sum(1) - 1.0) < 1e-5) and np.all(got[(legal[:, None] >> np.arange(7)) & 1 == 0] == 0)

    reqs = [(g, 0, 0) for g in [0, 3, 42, 43, 1000, 1001, 1002, 1003, 1004, 1005]]
    results = {}
    for name, noise in (("off", None), ("zero", (0.3, 0.0)), ("on", (0.3, 0.25)), ("alpha1", (1.0, 0.5))):
        s = DeviceSession(4, 20, 6.6, 0.01)
        s.set_games(reqs)
        if noise is not None:
            s.set_dirichlet(*noise)
        s.run(hash_eval_torch)
        results[name] = samples_by_game(s.drain_samples())
        s.close()
    plain, _ = O.self_play(reqs, 64, 20, 6.6, 0.01, "hash")
    assert results["off"] == results["zero"] == oracle_samples_by_game(plain)
    for name, noise in (("on", (0.3, 0.25)), ("alpha1", (1.0, 0.5))):
        want, _ = O.self_play(reqs, 64, 20, 6.6, 0.01, "hash", dirichlet=noise)
        assert results[name] == oracle_samples_by_game(want), name
        assert results[name] != results["off"]
    # both extensions together: the cache keeps the evaluator's RAW outputs, the noise goes into the
    # priors afterwards, so noisy games are the same with and without the cache
    s = DeviceSession(4, 20, 6.6, 0.01)
    s.set_games(reqs)
    s.set_dirichlet(0.3, 0.25)
    s.set_eval_cache(1 << 14)
    s.run(hash_eval_torch)
    both = samples_by_game(s.drain_samples())
    assert s.counters()["eval_cache_hits"] > 0
    s.close()
    assert both == results["on"]


@pytest.mark.parametrize("entries,max_sims", [(1 << 16, 0), (1024, 8)])   # roomy table; tiny table (constant eviction), long trips
def test_evaluation_cache_extension_keeps_every_sample(env, entries, max_sims):
    """c4_session_set_eval_cache (extension, off by default): positions whose evaluation is already
    in the table run their simulation without an evaluator row.  Under a deterministic evaluator
    nothing a game records may change: samples and the per-game work counters equal the oracle's,
    while the evaluator sees fewer rows."""
    DeviceSession, O, dev = env
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game

    reqs = [(gid, 0, 0) for gid in [0, 42, 43] + list(range(2000, 2045))]
    n_iter = 30
    evaluated = []

    def counting_eval(planes):
        evaluated.append(planes.shape[0])
        return hash_eval_torch(planes)

    out = {}
    for name, use_cache in (("off", False), ("on", True)):
        s = DeviceSession(16, n_iter, 6.6, 0.01, device=dev)
        s.set_games(reqs)
        if use_cache:
            s.set_eval_cache(entries, max_sims)
        steps = s.run(counting_eval)
        out[name] = (samples_by_game(s.drain_samples()), s.counters(), steps)
        s.close()
    ores, ost = O.self_play(reqs, 1 << 20, n_iter, 6.6, 0.01, "hash")
    want = oracle_samples_by_game(ores)
    (got_off, c_off, steps_off), (got_on, c_on, steps_on) = out["off"], out["on"]
    assert got_off == want and got_on == want
    for k in ("sims", "select_levels", "backup_nodes", "expansions", "moves", "games_done", "samples", "ref_skipped_sims"):
        assert c_on[k] == c_off[k], k
    assert c_off["eval_cache_probes"] == 0 and c_off["eval_cache_hits"] == 0
    assert 0 < c_on["eval_cache_hits"] <= c_on["eval_cache_probes"]
    assert steps_on < steps_off          # the same simulations in fewer evaluator passes


def test_evaluation_cache_is_refused_for_multi_model_sessions(env):
    DeviceSession, O, dev = env
    from c4a0_amd._lib import C4Error

    s = DeviceSession(8, 10, 6.6, 0.01, device=dev)
    s.bind_leaf_models()
    with pytest.raises(C4Error):
        s.set_eval_cache(4096)
    s.close()
    s = DeviceSession(8, 10, 6.6, 0.01, device=dev)
    s.set_eval_cache(4096)
    with pytest.raises(C4Error):
        s.bind_leaf_models()
    s.close()


def test_tail_compaction_moves_games_without_changing_them(env):
    """c4_session_compact: once the request list is exhausted the surviving games are moved into the
    lowest slots (state, arena, evaluator row) and the session narrows; every sample must still
    equal the oracle's, whatever the moments at which it is called."""
    DeviceSession, O, dev = env
    from tests.helpers import hash_eval_torch, oracle_samples_by_game, samples_by_game

    reqs = [(3000 + i, 0, 0) for i in range(90)]
    s = DeviceSession(64, 10, 6.6, 0.01, device=dev)
    s.set_games(reqs)
    s.bind()
    s.start()
    widths, steps = [], 0
    while s.counters()["games_done"] < len(reqs) and steps < 20000:
        s.evaluate(hash_eval_torch)
        s.step()
        steps += 1
        if steps % 7 == 0:
            act, rows = s.compact(8)
            widths.append(rows)
            assert act <= rows <= 64 and rows % 8 == 0 and rows == s.rows
            m, v, status = s.leaves()
            assert (status[:act] == 1).all() or rows == 64          # compacted: the games sit in the first slots
    got = samples_by_game(s.drain_samples())
    c = s.counters()
    # a new list of games restores the full width
    s.set_games(reqs[:3])
    assert s.rows == 64
    s.close()
    assert widths[0] == 64 and widths[-1] == 8 and sorted(widths, reverse=True) == widths   # no-op while requests are queued, then it narrows
    ores, ost = O.self_play(reqs, 1 << 20, 10, 6.6, 0.01, "hash")
    assert got == oracle_samples_by_game(ores)
    assert c["sims"] + c["ref_skipped_sims"] == ost["sims"] and c["moves"] == ost["moves"] and c["games_done"] == 90
