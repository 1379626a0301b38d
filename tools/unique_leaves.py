import sys, os, ctypes as C, torch
sys.path.insert(0, os.getcwd())
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
from c4a0_amd.session import DeviceSession
from c4a0_amd._lib import check
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), dev, dtype=torch.bfloat16)
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = DeviceSession(G, 100, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
s.set_games([(i, 0, 0) for i in range(G * 6)])
s.bind(); s.start()
keys = torch.zeros(G, dtype=torch.int64, device=dev)
fr = []
for step in range(6000):
    s.evaluate(net); s.step()
    if step >= 3000 and step % 100 == 0:
        check(s.L.c4_session_leaf_keys(s._h, C.c_void_p(keys.data_ptr())))
        k = keys[keys >= 0]
        fr.append((int(torch.unique(k).numel()), int(k.numel())))
u = sum(a for a, b in fr); t = sum(b for a, b in fr)
print("games", G, "steady-state unique leaves / active slots:", u, t, round(u / t, 4), "per-step min/max", min(a / b for a, b in fr), max(a / b for a, b in fr))
