import os, sys, time, torch
sys.path.insert(0, ".")
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(1700)]
for k in (1, 2):
    c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net, concurrent_sessions=k)
ref = None
for k in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else '1,2,1,2,1,2').split(',')]:
    if True:
        st = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, evaluator=net, concurrent_sessions=k, stats=st)
        t1 = time.perf_counter()
        recs, _ = res.to_records(); t2 = time.perf_counter()
        ref = recs if ref is None else ref
        print(f"sessions={k}: play {t1 - t0:.3f} s + records {t2 - t1:.3f} s = {1700 / (t2 - t0):.0f} games/s, {st['steps']} steps, same: {recs.tobytes() == ref.tobytes()}", flush=True)
