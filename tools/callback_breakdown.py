#!/usr/bin/env python3
"""Where a step of the numpy-callback mode goes: the user's callback itself (H2D, forward, 3 x D2H -- not ours
to change) against everything play_games does around it (unique leaves, the PCIe hops, the fan-out, the step
kernel).  usage: callback_breakdown.py [games] [n_mcts] [blocks] [channels] [policy_layers] [value_layers]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

a = [int(x) for x in sys.argv[1:]] + [None] * 6
games, n_mcts = a[0] or 1700, a[1] or 1400
cfg = ModelConfig(a[2] or 1, a[3] or 32, a[4] or 4, a[5] or 2)
cap = int(os.environ.get("CAP", "2000"))
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(cfg), dev)
spent = {"cb": 0.0, "calls": 0, "rows": 0}

def cb(_model_id, x):   # the shape of ConnectFourNet.forward_numpy (nn.py:119-130)
    t = time.perf_counter()
    with torch.no_grad():
        lp, q = net(torch.from_numpy(x).to(dev))
        lp, q = lp.cpu().numpy(), q.cpu().numpy()
    out = np.ascontiguousarray(lp), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])
    spent["cb"] += time.perf_counter() - t
    spent["calls"] += 1
    spent["rows"] += x.shape[0]
    return out

reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(games)]
for rep in range(2):
    spent.update(cb=0.0, calls=0, rows=0)
    stats = {}
    t0 = time.perf_counter()
    c4a0_amd.play_games(reqs, cap, n_mcts, 6.6, 0.01, cb, stats=stats)
    dt = time.perf_counter() - t0
    steps = stats["steps"]
    print(f"run {rep}: {games} games n={n_mcts}: {dt:.2f} s = {games / dt:.0f} games/s, {steps} steps, {dt / steps * 1e6:.0f} us/step; "
          f"callback {spent['cb'] / spent['calls'] * 1e6:.0f} us/call x {spent['calls']} calls ({spent['rows'] / spent['calls']:.0f} rows/call) = "
          f"{spent['cb'] / dt:.0%} of the wall; around it {(dt - spent['cb']) / steps * 1e6:.0f} us/step")
