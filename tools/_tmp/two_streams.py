import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from c4a0_amd.session import DeviceSession
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(1337)
model = ConnectFourNet(ModelConfig(4, 32, 4, 2))
def make(G, base):
    net = InferenceNet(model, dev)
    s = DeviceSession(G, 100, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games([(base + i, 0, 0) for i in range(G * 200)])
    s.bind(); s.start()
    g = s.capture_steps(net, 8)
    return s, g, net
def run(parts, preroll_graphs, timed_graphs):
    streams = [torch.cuda.Stream(device=dev) for _ in parts]
    def go(n):
        for _ in range(n):
            for (s, g, _), st in zip(parts, streams):
                with torch.cuda.stream(st):
                    g.replay()
            # bound the host run-ahead
        for st in streams: st.synchronize()
    go(preroll_graphs)
    d0 = sum(s.counters()["games_done"] for s, _, _ in parts)
    sims0 = sum(s.counters()["sims"] for s, _, _ in parts)
    t0 = time.perf_counter()
    go(timed_graphs)
    dt = time.perf_counter() - t0
    d1 = sum(s.counters()["games_done"] for s, _, _ in parts)
    sims1 = sum(s.counters()["sims"] for s, _, _ in parts)
    return (d1 - d0) / dt, (sims1 - sims0) / dt, dt / (timed_graphs * 8) * 1e6
for cfg in ([4096], [2048, 2048], [2048, 2048], [1408, 1344, 1344], [2560, 1536], [3072, 1024], [4096]):
    parts = [make(G, 10_000_000 * i) for i, G in enumerate(cfg)]
    gps, sps, us = run(parts, 400, 400)
    print(cfg, f"{gps:.0f} games/s  {sps / 1e6:.2f} M sims/s  {us:.1f} us per step-round")
    for s, _, _ in parts: s.close()
    del parts
