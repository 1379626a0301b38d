#!/usr/bin/env python3
"""Randomised parity soak: many small self-play jobs with random shapes and options, HIP vs the oracle,
every sample of every game bit for bit (hash evaluator).  Run on a GPU box:

    python tools/fuzz_parity.py [seconds]
"""
import os, sys, time, random
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd._lib import C4Error
from c4a0_amd.session import DeviceSession
from oracle import c4oracle as O
from tests.helpers import GraphSafeHashEval, hash_eval_torch, oracle_samples_by_game, samples_by_game

import c4a0_amd
from tests.helpers import hash_eval_np

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 20260002)
rng_net = random.Random((int(sys.argv[2]) if len(sys.argv) > 2 else 20260002) * 7919 + 64)   # network shapes (channels, blocks): their own generator
t0, n_jobs, n_games_total, n_errs, n_cb_jobs = time.time(), 0, 0, 0, 0
n_reclaimed_jobs = 0
n_dirichlet_jobs = 0


def player(model_id, x):      # every "model" prefers other columns (the oracle calls the same function)
    lp, qp, qn = hash_eval_np(model_id, x)
    return np.ascontiguousarray(np.roll(lp, int(model_id % 7), axis=1)), qp, qn


def callback_job():
    """play_games through the numpy callback (unique (model, position) batches built on the device,
    c4_session_unique_leaves / c4_session_scatter_outputs), one or several models, against the oracle driven by the
    same callback."""
    global n_jobs, n_games_total, n_errs, n_cb_jobs
    n_games = rng.choice([1, 2, 5, 9, 24, 70])
    n_iter = rng.choice([2, 3, 5, 10, 25, 60])
    cap = rng.choice([1, 3, 8, 64, 4096])
    resident = rng.choice([None, 1, 3, 8, 33])
    models = rng.choice([[0], [5], [0, 1], [3, 2**63 + 1, 2**64 - 1], [7, 8, 9, 10]])
    c_expl, c_ply = rng.choice([0.5, 1.4, 6.6]), rng.choice([0.0, 0.01])
    reqs = [(1000 + i, rng.choice(models), rng.choice(models)) for i in range(n_games)]
    cfg = dict(callback=True, n_games=n_games, n_iter=n_iter, cap=cap, resident=resident, models=models, c_expl=c_expl, c_ply=c_ply)
    batches = []

    def cb(model_id, x):
        assert x.dtype == np.float32 and x.shape[1:] == (2, 6, 7) and 0 < x.shape[0] <= cap and x.flags["C_CONTIGUOUS"], cfg
        assert len({x[i].tobytes() for i in range(x.shape[0])}) == x.shape[0], cfg
        batches.append(x)         # kept: nothing handed over may ever be overwritten
        return player(model_id, x)

    dev_err = ora_err = None
    try:
        got = c4a0_amd.play_games([c4a0_amd.GameMetadata(*r) for r in reqs], cap, n_iter, c_expl, c_ply, cb, resident_games=resident)
    except C4Error as e:
        dev_err = e
    kept = [b.copy() for b in batches]
    try:
        want, _ = O.self_play(reqs, cap, n_iter, c_expl, c_ply, player)
    except RuntimeError as e:
        ora_err = e
    assert (dev_err is None) == (ora_err is None), (cfg, dev_err, ora_err)
    assert all(np.array_equal(a, b) for a, b in zip(batches, kept)), cfg
    if dev_err is not None:
        n_errs += 1
        return
    by_game = {r.metadata.game_id: [(x.mask, x.value, x.policy.tobytes(), x.q_penalty.tobytes(), x.q_no_penalty.tobytes()) for x in r.samples]
               for r in got.results}
    assert by_game == oracle_samples_by_game(want), cfg
    assert [(r.metadata.game_id, r.metadata.player0_id, r.metadata.player1_id) for r in got.results] == reqs, cfg
    n_jobs += 1
    n_cb_jobs += 1
    n_games_total += n_games


_nets = {}
n_fused_jobs = 0


def fused_job():
    """The bf16 network in device mode: play_games (HIP graphs; the output layers inside the step's launch,
    c4_session_step_head_out; two sessions from 2 048 slots) against an eager DeviceSession.run (stand-alone output kernel and
    step kernel, per-launch timing on) -- byte-identical records for random widths incl. non-multiples of 16 and refills."""
    global n_jobs, n_games_total, n_fused_jobs
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from c4a0_amd.results import GameMetadata, results_from_records
    heads = rng.choice([(2, 2), (4, 2), (1, 1), (3, 1)])
    heads = heads + (rng_net.choice([32, 32, 64]),)     # (its own generator: the job sequence of a seed stays what it was)
    if heads not in _nets:
        torch.manual_seed(hash(heads) & 0xFFFF)
        _nets[heads] = InferenceNet(ConnectFourNet(ModelConfig(rng_net.choice([1, 2]), heads[2], *heads[:2])), torch.device("cuda:0"), dtype=torch.bfloat16)
    net = _nets[heads]
    n_games = rng.choice([1, 5, 16, 17, 40, 130])
    n_slots = rng.choice([1, 7, 8, 15, 16, 17, 31, 33, 100])
    n_iter = rng.choice([2, 5, 12, 30])
    reqs = [(5000 + 3 * i, 0, 0) for i in range(n_games)]
    cfg = dict(fused=True, heads=heads, n_games=n_games, n_slots=n_slots, n_iter=n_iter)
    got = c4a0_amd.play_games([c4a0_amd.GameMetadata(*r) for r in reqs], 64, n_iter, 6.6, 0.01, evaluator=net, resident_games=n_slots,
                              concurrent_sessions=rng.choice([1, 1, 2]), host_loop="python")   # (native_job covers the library's own loop)
    s = DeviceSession(min(n_slots, n_games), n_iter, 6.6, 0.01, planes_dtype=torch.bfloat16)
    s.set_games(reqs)
    s.run(net)                                   # eager: evaluate() + step(), the two stand-alone kernels
    want = results_from_records([GameMetadata(*r) for r in reqs], s.drain_samples(), s.sample_counts())
    s.close()
    assert got.to_records()[0].tobytes() == want.to_records()[0].tobytes(), cfg
    n_jobs += 1
    n_fused_jobs += 1
    n_games_total += n_games


n_native_jobs = 0


def native_job():
    """The library's own host loop (c4_play_games_bf16: sessions, paired graph, narrowing, merged hand-over in C++) against an eager
    DeviceSession.run of the same games (stand-alone kernels launched from Python): byte-identical records for random networks,
    widths, one / two sessions, graph lengths, and with the extensions / reclaimed arenas."""
    global n_jobs, n_games_total, n_native_jobs
    from c4a0_amd.native import play_games_native
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from c4a0_amd.results import GameMetadata, results_from_records
    heads = rng.choice([(2, 2), (4, 2), (3, 3)])
    key = ("native",) + heads + (rng_net.choice([32, 32, 64]),)
    if key not in _nets:
        torch.manual_seed(hash(key) & 0xFFFF)
        _nets[key] = InferenceNet(ConnectFourNet(ModelConfig(rng_net.choice([1, 2]), key[3], *heads)), torch.device("cuda:0"), dtype=torch.bfloat16)
    net = _nets[key]
    n_games = rng.choice([1, 5, 16, 17, 40, 130, 600])
    n_slots = rng.choice([1, 7, 8, 15, 16, 33, 100, 300])
    n_iter = rng.choice([2, 5, 12, 30])
    sessions = rng.choice([1, 1, 2])
    dirichlet = rng.choice([None, None, (0.3, 0.25)])
    cache = rng.choice([0, 0, 4096])
    rperiod = rng.choice([1, 2, 5])
    rkw = dict(reclaim=True, reclaim_period=rperiod, blocks_per_slot=2 * (n_iter + 16 + 2 * (16 * rperiod + 16) + rng.choice([0, 3, 40]))) if rng.random() < 0.3 else {}
    reqs = [(9000 + 5 * i, 0, 0) for i in range(n_games)]
    cfg = dict(native=True, heads=heads, n_games=n_games, n_slots=n_slots, n_iter=n_iter, sessions=sessions, dirichlet=dirichlet, cache=cache, reclaim=rkw)
    got = play_games_native([GameMetadata(*r) for r in reqs], 64, n_iter, 6.6, 0.01, net, resident_games=n_slots, concurrent_sessions=sessions,
                            steps_per_graph=rng.choice([0, 1, 3, 8]), tail_steps_per_graph=rng.choice([0, 1, 4]), dirichlet=dirichlet,
                            eval_cache_entries=cache, **rkw)
    s = DeviceSession(min(n_slots, n_games), n_iter, 6.6, 0.01, planes_dtype=torch.bfloat16)
    s.set_games(reqs)
    if dirichlet:
        s.set_dirichlet(*dirichlet)
    s.run(net)                                   # eager: evaluate() + step(), the two stand-alone kernels, no cache
    want = results_from_records([GameMetadata(*r) for r in reqs], s.drain_samples(), s.sample_counts())
    s.close()
    assert got.to_records()[0].tobytes() == want.to_records()[0].tobytes(), cfg
    n_jobs += 1
    n_native_jobs += 1
    n_games_total += n_games


while time.time() - t0 < budget:
    r = rng.random()
    if r < 0.25 and "FUZZ_TREE_ONLY" not in os.environ:
        callback_job()
        continue
    if r < 0.45 and "FUZZ_TREE_ONLY" not in os.environ:
        fused_job()
        continue
    if r < 0.55 and "FUZZ_TREE_ONLY" not in os.environ:
        native_job()
        continue
    n_games = rng.choice([1, 2, 3, 7, 8, 9, 17, 40, 100])
    n_slots = rng.choice([1, 2, 7, 8, 9, 16, 33])
    n_iter = rng.choice([1, 2, 3, 5, 10, 25, 60, 150])
    c_expl = rng.choice([0.0, 0.5, 1.4, 4.0, 6.6, 25.0])
    c_ply = rng.choice([0.0, 0.001, 0.01, 0.02])
    dirichlet = rng.choice([None, None, (0.3, 0.25), (1.0, 0.5), (0.05, 0.1)])
    if "FUZZ_DIRICHLET_SHARE" in os.environ:   # (FUZZ_DIRICHLET_SHARE=1: every tree job with root noise)
        dirichlet = rng.choice([(0.3, 0.25), (1.0, 0.5), (0.05, 0.1)]) if rng.random() < float(os.environ["FUZZ_DIRICHLET_SHARE"]) else None
    cache = rng.choice([0, 0, 1024, 1 << 16])
    one_sim = rng.random() < 0.2
    graph = rng.random() < 0.4
    ids = [rng.choice([0, 1, 42, 43, 2**64 - 1, rng.getrandbits(64), rng.randrange(1000)]) for _ in range(n_games)]
    ids = list(dict.fromkeys(ids))          # the oracle's result dict is keyed by game id
    reqs = [(g, 0, 0) for g in ids]
    # a third of the jobs on a RECLAIMED arena (C4_FLAG_RECLAIM) with halves near the smallest the library accepts and a look at the
    # arenas every 1-7 launches: the live subtree is copied into the other half several times per game
    reclaim = rng.random() < float(os.environ.get("FUZZ_RECLAIM_SHARE", "0.33"))   # (FUZZ_RECLAIM_SHARE=1: every tree job on a reclaimed arena)
    rperiod = rng.choice([1, 1, 2, 3, 7])
    rkw = dict(reclaim=True, reclaim_period=rperiod, blocks_per_slot=2 * (n_iter + 10 + 2 * (4 * rperiod + 16) + rng.choice([0, 0, 1, 7, 50]))) if reclaim else {}
    s = DeviceSession(n_slots, n_iter, c_expl, c_ply, planes_dtype=rng.choice([torch.float32, torch.bfloat16]), one_sim_per_step=one_sim, **rkw)
    s.set_games(reqs)
    if dirichlet:
        s.set_dirichlet(*dirichlet)
    if cache:
        s.set_eval_cache(cache)
    cfg = dict(n_games=len(ids), n_slots=n_slots, n_iter=n_iter, c_expl=c_expl, c_ply=c_ply, dirichlet=dirichlet, cache=cache, one_sim=one_sim, graph=graph, reclaim=rkw)
    dev_err = None
    try:
        if graph:
            s.run(GraphSafeHashEval(), steps_per_graph=rng.choice([1, 4, 8]))
        else:
            s.run(hash_eval_torch)
        got = samples_by_game(s.drain_samples())
        ctr = s.counters()
    except C4Error as e:       # where the reference panics (e.g. tiny n: a uniform root policy samples a full column, mcts.rs:196-200)
        dev_err = e
    s.close()
    try:
        want, _ = O.self_play(reqs, 64, n_iter, c_expl, c_ply, "hash", dirichlet=dirichlet or (0.0, 0.0))
        ora_err = None
    except RuntimeError as e:
        ora_err = e
    assert (dev_err is None) == (ora_err is None), (cfg, dev_err, ora_err)
    if dev_err is not None:
        n_errs += 1
        continue
    assert ctr["error"] == 0 and ctr["games_done"] == len(ids), (cfg, ctr)
    n_reclaimed_jobs += 1 if ctr["reclaim_passes"] else 0
    n_dirichlet_jobs += 1 if dirichlet else 0
    assert got == oracle_samples_by_game(want), cfg
    n_jobs += 1
    n_games_total += len(ids)
print(f"fuzz parity ok: {n_jobs} jobs ({n_cb_jobs} of them through the numpy callback, one or several models; {n_fused_jobs} with the bf16 network, fused graph path vs eager; {n_native_jobs} through the native host loop c4_play_games_bf16 vs eager; {n_reclaimed_jobs} on arenas reclaimed during play; {n_dirichlet_jobs} with Dirichlet noise), {n_games_total} games in {time.time() - t0:.0f} s; "
      f"{n_errs} more jobs ended in the same panic on both sides")
