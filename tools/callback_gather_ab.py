"""Same-box A/B of the callback mode's scatter + step as one launch (c4_session_step_gather) on the reference's default job."""
import sys, time, torch
sys.path.insert(0, ".")
import c4a0_amd
from c4a0_amd.api import _CallbackEvaluator
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
cb = lambda _m, x: net.forward_numpy(x)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n)]
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, cb)
ref = None
for rep in range(2):
    for gather in (False, True):
        _CallbackEvaluator.gather_step = gather
        t0 = time.perf_counter()
        res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, cb)
        recs, _ = res.to_records(); dt = time.perf_counter() - t0
        ref = recs if ref is None else ref
        print(f"gather_step={gather}: {n / dt:.0f} games/s ({dt:.3f} s) identical={recs.tobytes() == ref.tobytes()}", flush=True)
