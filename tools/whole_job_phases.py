import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
from c4a0_amd.session import DeviceSession
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(1700)]
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net)
cap = {"n": 0, "t": 0.0}
_orig = DeviceSession.capture_steps
def _timed(self, *a, **k):
    torch.cuda.synchronize(); t = time.perf_counter()
    g = _orig(self, *a, **k)
    torch.cuda.synchronize(); cap["t"] += time.perf_counter() - t; cap["n"] += 1
    return g
DeviceSession.capture_steps = _timed
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = DeviceSession(1700, 1400, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.set_games([(r.game_id, 0, 0) for r in reqs])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    net.latency_mode = True
    steps = s.run(net, steps_per_graph=8)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    recs = s.pack_samples_device(); cnt = s.sample_counts()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    s.close()
    torch.cuda.synchronize(); t5 = time.perf_counter()
    print(f"captures {cap['n']} in {cap['t']:.3f} s; rows at end {s.rows}")
    cap.update(n=0, t=0.0)
    print(f"create {t1-t0:.3f}  set_games {t2-t1:.3f}  run {t3-t2:.3f} ({steps} steps, {(t3-t2)/steps*1e6:.1f} us/step)  pack {t4-t3:.3f}  close {t5-t4:.3f}  total {t5-t0:.3f}")
st = {}
t0 = time.perf_counter(); res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, evaluator=net, stats=st); a = res.to_records(); t1 = time.perf_counter()
print(f"play_games + to_records {t1-t0:.3f} s")
