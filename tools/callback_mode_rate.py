#!/usr/bin/env python3
"""Rate of the reference-compatible numpy-callback mode of play_games (host round trip per step):
what an unmodified c4a0 training loop gets from the GPU tree with its own forward_numpy-style callback."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), dev)

def cb(_model_id, x):   # the shape of ConnectFourNet.forward_numpy (nn.py:119-130): H2D, forward, 3 x D2H
    with torch.no_grad():
        lp, q = net(torch.from_numpy(x).to(dev))
        lp, q = lp.cpu().numpy(), q.cpu().numpy()
    return np.ascontiguousarray(lp), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n)]
stats = {}
t0 = time.perf_counter()
res = c4a0_amd.play_games(reqs, 4096, 100, 6.6, 0.01, cb, stats=stats)
dt = time.perf_counter() - t0
print(f"callback mode: {n} games, n_mcts=100: {dt:.1f} s = {n / dt:.0f} games/s, {stats['sims'] / dt / 1e6:.2f} M sims/s, {stats['steps']} steps")
t0 = time.perf_counter()
c4a0_amd.play_games(reqs, 4096, 100, 6.6, 0.01, lambda _m, x: net.forward_numpy(x), stats=stats)   # a caller whose model IS an InferenceNet (nn.py:119-130 entry point)
dt = time.perf_counter() - t0
print(f"callback mode, callback = InferenceNet.forward_numpy (same fp32 weights; one HIP-graph replay per call): {dt:.1f} s = {n / dt:.0f} games/s")
t0 = time.perf_counter()
res2 = c4a0_amd.play_games(reqs, 4096, 100, 6.6, 0.01, evaluator=net, stats=stats)
dt = time.perf_counter() - t0
print(f"device mode (default resident games = {stats['n_slots']}): {n} games: {dt:.2f} s = {n / dt:.0f} games/s, {stats['steps']} steps")
a = res.to_arrays()
print("device-mode samples identical to the callback run:", all(np.array_equal(x, y) for x, y in zip(a, res2.to_arrays())))
for name, kw in (("callback mode + evaluation cache (extension)", dict(py_eval_pos_cb=cb)), ("device mode + evaluation cache (extension)", dict(evaluator=net))):
    t0 = time.perf_counter()
    r = c4a0_amd.play_games(reqs, 4096, 100, 6.6, 0.01, eval_cache_entries=1 << 24, stats=stats, **kw)
    dt = time.perf_counter() - t0
    same = all(np.array_equal(x, y) for x, y in zip(a, r.to_arrays()))
    print(f"{name}: {n} games: {dt:.2f} s = {n / dt:.0f} games/s, {stats['steps']} steps, hit rate "
          f"{stats['eval_cache_hits'] / max(1, stats['eval_cache_probes']):.2f}, samples identical to the plain callback run: {same}")
