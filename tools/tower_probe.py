#!/usr/bin/env python3
"""Runs the HIP conv tower alone (for rocprofv3 --pmc passes and timing)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
dev = torch.device("cuda:0")
ch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
tower_config = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # c4_conv_tower_bf16's workgroup shape, 0 = automatic
torch.manual_seed(0)
net = InferenceNet(ConnectFourNet(ModelConfig(blocks, ch, 4, 2)), dev, tower_config=tower_config)
x = (torch.rand(n, 2, 6, 7, device=dev) > 0.7).to(torch.bfloat16)
for _ in range(5):
    net.tower(x)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(50):
    net.tower(x)
b.record()
torch.cuda.synchronize()
print(f"tower C={ch} blocks={blocks} n={n}: {a.elapsed_time(b) / 50 * 1e3:.1f} us per call")
