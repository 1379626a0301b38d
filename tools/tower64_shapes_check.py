import sys, os, torch
sys.path.insert(0, os.getcwd())
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ConnectFourNet(ModelConfig(8, 64, 4, 2))
x = (torch.rand(3000, 2, 6, 7, device=dev) > 0.7).to(torch.bfloat16)
outs = []
for cfg in (0, 2, 3):
    net = InferenceNet(model, dev, tower_config=cfg)
    outs.append(net.tower(x).clone())
print("config 2 == config 0:", torch.equal(outs[0], outs[1]), " config 3 == config 0:", torch.equal(outs[0], outs[2]))
