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

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = random.Random(20260002)
t0, n_jobs, n_games_total, n_errs = time.time(), 0, 0, 0
while time.time() - t0 < budget:
    n_games = rng.choice([1, 2, 3, 7, 8, 9, 17, 40, 100])
    n_slots = rng.choice([1, 2, 7, 8, 9, 16, 33])
    n_iter = rng.choice([1, 2, 3, 5, 10, 25, 60, 150])
    c_expl = rng.choice([0.0, 0.5, 1.4, 4.0, 6.6, 25.0])
    c_ply = rng.choice([0.0, 0.001, 0.01, 0.02])
    dirichlet = rng.choice([None, None, (0.3, 0.25), (1.0, 0.5), (0.05, 0.1)])
    cache = rng.choice([0, 0, 1024, 1 << 16])
    one_sim = rng.random() < 0.2
    graph = rng.random() < 0.4
    ids = [rng.choice([0, 1, 42, 43, 2**64 - 1, rng.getrandbits(64), rng.randrange(1000)]) for _ in range(n_games)]
    ids = list(dict.fromkeys(ids))          # the oracle's result dict is keyed by game id
    reqs = [(g, 0, 0) for g in ids]
    s = DeviceSession(n_slots, n_iter, c_expl, c_ply, planes_dtype=rng.choice([torch.float32, torch.bfloat16]), one_sim_per_step=one_sim)
    s.set_games(reqs)
    if dirichlet:
        s.set_dirichlet(*dirichlet)
    if cache:
        s.set_eval_cache(cache)
    cfg = dict(n_games=len(ids), n_slots=n_slots, n_iter=n_iter, c_expl=c_expl, c_ply=c_ply, dirichlet=dirichlet, cache=cache, one_sim=one_sim, graph=graph)
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
    assert got == oracle_samples_by_game(want), cfg
    n_jobs += 1
    n_games_total += len(ids)
print(f"fuzz parity ok: {n_jobs} jobs, {n_games_total} games in {time.time() - t0:.0f} s; {n_errs} more jobs ended in the same panic on both sides")
