#!/usr/bin/env python3
"""Where the reference's default job spends its time OUTSIDE the rounds: session creation (tree arena), set_games, start, graph
capture, the first replay, the sample hand-over -- first call (fresh arena) and second call (arena kept by the library)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
from c4a0_amd.session import DeviceSession

dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(1700)]
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net)
for call in range(3):
    t = [time.perf_counter()]
    def lap():
        torch.cuda.synchronize(); t.append(time.perf_counter())
    s = DeviceSession(1700, 1400, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16); lap()
    s.set_games([(r.game_id, 0, 0) for r in reqs]); lap()
    net.latency_mode = True
    s.bind(); s.start(); lap()
    g = s.capture_steps(net, 32); lap()
    g.replay(); lap()
    g.replay(); lap()
    s.close(); lap()
    names = ["create", "set_games", "bind + start", "capture 32 rounds", "first replay", "second replay", "close"]
    print(f"call {call}: " + "  ".join(f"{n} {1e3 * (b - a):.1f} ms" for n, a, b in zip(names, t, t[1:])), flush=True)
t0 = time.perf_counter(); st = {}
res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, evaluator=net, stats=st); t1 = time.perf_counter()
recs, _ = res.to_records(); t2 = time.perf_counter()
print(f"whole job {t1 - t0:.3f} s + to_records {1e3 * (t2 - t1):.1f} ms; {st['steps']} rounds")
