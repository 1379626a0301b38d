#!/usr/bin/env python3
"""Whole-job rate of play_games(evaluator=) as a function of concurrent_sessions, for jobs whose batch is
small enough that one session's kernel chain is latency-bound (the reference's default job: 1 700 games,
n_mcts_iterations = 1 400, 1 x 32 network)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

n_games = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
blocks, ch = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1, 32)
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(blocks, ch, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n_games)]
ref = None
for k in [int(x) for x in os.environ.get("SESSIONS", "1,2,3,4").split(",")]:
    st = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = c4a0_amd.play_games(reqs, 2000, n_iter, 6.6, 0.01, evaluator=net, concurrent_sessions=k, stats=st)
    recs, _ = res.to_records()
    dt = time.perf_counter() - t0
    ref = recs if ref is None else ref
    print(f"sessions={k}: {dt:.2f} s = {n_games / dt:.0f} games/s, {st['sims'] / dt / 1e6:.1f} M sims/s, {st['steps']} steps, same samples as 1 session: {recs.tobytes() == ref.tobytes()}", flush=True)
