import sys, time, json, numpy as np, torch
sys.path.insert(0, ".")
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
from c4a0_amd.session import DeviceSession
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(1700)]
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net)
ref = None
for rep in range(2):
    for fused in (False, True):
        InferenceNet.wide_tiles_r5 = fused         # (this run: the A/B is round 5's tiles for the 2F-wide layer at 1 025-1 728 rows)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, evaluator=net)
        recs, _ = res.to_records(); dt = time.perf_counter() - t0
        ref = recs if ref is None else ref
        print(f"wide_tiles_r5={fused}: {1700 / dt:.0f} games/s ({dt:.3f} s) identical={recs.tobytes() == ref.tobytes()}", flush=True)
