#!/usr/bin/env python3
"""The evaluator alone, eagerly, for rocprofv3 passes (kernel trace or --pmc):
    python3 tools/evaluator_probe.py [M] [gemm backend: hip|hipblaslt] [gemm config] [blocks] [channels] [reps]
Every launch is one kernel of: c4_conv_tower_kernel, the three hidden-layer GEMMs, c4_head_out_mfma_kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
backend = sys.argv[2] if len(sys.argv) > 2 else "hip"
cfg = sys.argv[3] if len(sys.argv) > 3 else "0"
blocks = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ch = int(sys.argv[5]) if len(sys.argv) > 5 else 32
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 30
dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(blocks, ch, 4, 2)), dev, dtype=torch.bfloat16, gemm=backend, gemm_config=sys.argv[3] if len(sys.argv) > 3 else 0)
x = (torch.rand(M, 2, 6, 7, device=dev) > 0.7).to(torch.bfloat16)
lp = torch.empty((M, 7), dtype=torch.float32, device=dev)
q = torch.empty((M, 2), dtype=torch.float32, device=dev)
for _ in range(reps):
    net(x, out_logprobs=lp, out_q=q)
torch.cuda.synchronize()
print("done", M, backend, cfg, float(lp.sum()))
