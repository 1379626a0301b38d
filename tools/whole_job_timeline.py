"""The reference's default job (1 700 games, n = 1 400, 1x32 net) as a timeline: games done, resident rows and the
time per lock-step round at every progress poll -- how much of the job is its tail.
    python tools/whole_job_timeline.py [games=1700] [n_mcts=1400]"""
import sys, time, torch
sys.path.insert(0, ".")
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
from c4a0_amd.session import DeviceSession

n_games = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 1400
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n_games)]
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net)
log = []
poll0 = DeviceSession.poll
def poll(self):
    r = poll0(self)
    log.append((time.perf_counter(), r[0], self.rows))
    return r
DeviceSession.poll = poll
torch.cuda.synchronize(); t0 = time.perf_counter()
res = c4a0_amd.play_games(reqs, 2000, n_iter, 6.6, 0.01, evaluator=net)
dt = time.perf_counter() - t0
print(f"{n_games / dt:.0f} games/s ({dt:.3f} s), {len(log)} polls")
# the host runs up to two graph replays ahead of the device, so poll times are enqueue times: smooth over 16 polls
step = max(1, len(log) // 40)
prev = (t0, 0, 0)
for i in range(step - 1, len(log), step):
    t, done, rows = log[i]
    print(f"poll {i + 1:5d}  t = {1e3 * (t - t0):8.1f} ms  done {done:5d}  live {n_games - done:5d}  rows {rows:5d}  "
          f"{1e3 * (t - prev[0]) / step:7.3f} ms per poll")
    prev = log[i]
