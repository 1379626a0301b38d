export TMPDIR=/tmp; mkdir -p gpurun_out/wj
cat > /tmp/wj_dev.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.getcwd())
import c4a0_amd
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(1700)]
st = {}
t0 = time.perf_counter()
res = c4a0_amd.play_games(reqs, 2000, 1400, 6.6, 0.01, evaluator=net, stats=st)
dt = time.perf_counter() - t0
print("games/s", 1700 / dt, "steps", st["steps"], "us/step", dt / st["steps"] * 1e6)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wj/stats -- python3 /tmp/wj_dev.py > gpurun_out/wj/run.log 2>&1
cp $(find gpurun_out/wj/stats -name '*kernel_stats.csv' | head -1) gpurun_out/wj/kernel_stats.csv; rm -rf gpurun_out/wj/stats
tail -2 gpurun_out/wj/run.log; cut -c1-150 gpurun_out/wj/kernel_stats.csv | head -12
