import sys, time, torch
sys.path.insert(0, ".")
import c4a0_amd, c4a0_amd.session as S
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), dev, dtype=torch.bfloat16)
n = 16384
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n)]
c4a0_amd.play_games(reqs[:4096], 4096, 100, 6.6, 0.01, evaluator=net)
orig = S.run_sessions
ref = None
for spg in (8, 16, 8, 16, 8, 16, 8, 16, 32, 32):
    def patched(sessions, evaluator, steps_per_graph=8, **kw):
        return orig(sessions, evaluator, steps_per_graph=spg, **kw)
    S.run_sessions = patched
    st = {}
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = c4a0_amd.play_games(reqs, 4096, 100, 6.6, 0.01, evaluator=net, stats=st)
    dt = time.perf_counter() - t0
    recs, _ = res.to_records()
    ref = recs if ref is None else ref
    print(f"steps_per_graph {spg}: {dt:.3f} s = {n / dt:.0f} games/s, {st['steps']} rounds, {1e6 * dt / st['steps']:.1f} us per round, same {recs.tobytes() == ref.tobytes()}", flush=True)
