#!/usr/bin/env python3
"""Every tile configuration of c4_linear_bf16, ALONE, over a grid of row counts: the three fastest per (rows, layer) beside what
InferenceNet's latency mode asks for (tools/gemm_probe.py does the timing).   python tools/gemm_sweep.py [rows ...]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rows = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 512, 640, 768, 896, 1024, 1152, 1280, 1408, 1536, 1664, 1728]
env = dict(os.environ, PROBE_ONLY="alone", PROBE_CFGS=",".join(str(c) for c in range(1, 60)))
import torch
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), torch.device("cuda:0"))
for m in rows:
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "gemm_probe.py"), str(m)], env=env, capture_output=True, text=True, timeout=600).stdout
    for line in out.splitlines():
        mm = re.match(r"M(\d+) N(\d+) alone: (.*)", line)
        if not mm:
            continue
        n = int(mm.group(2))
        t = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.]+)us", mm.group(3))}
        hip = sorted(((v, k) for k, v in t.items() if k.startswith("hip") and k != "hipblaslt"))
        chosen = net._alone_config(m, n, 1344, True) or 0
        auto = t.get(f"hip{chosen}") if chosen else None
        print(f"rows {m:5d} N {n}: best " + "  ".join(f"{k[3:]}={v:.1f}" for v, k in hip[:4]) + f" | latency mode asks for {chosen or 'auto'}" + (f" = {auto:.1f} us" if auto else "") + f" | hipblaslt {t['hipblaslt']:.1f}", flush=True)
