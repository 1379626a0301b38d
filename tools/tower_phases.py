#!/usr/bin/env python3
"""Where a workgroup of the 32-channel tower spends its time (diagnostic build: python c4a0_amd/csrc/build.py --diag).
    python tools/tower_phases.py [boards=2048] [blocks=4] [tower_config=0] [channels=32]
64 channels (the streamed kernel): per layer k-loop | epilogue | hand-over between the two wavefronts of a pair."""
import ctypes as C, os, sys
os.environ.setdefault("C4A0_HIP_LIB", "libc4a0_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from c4a0_amd import _lib
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda:0")
L = _lib.lib()
L.c4_debug_tower_phases.restype = C.c_int
L.c4_debug_tower_phases.argtypes = [C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
torch.manual_seed(0)
net = InferenceNet(ConnectFourNet(ModelConfig(blocks, ch, 4, 2)), dev, tower_config=cfg)
x = (torch.rand(n, 2, 6, 7, device=dev) > 0.7).to(torch.bfloat16)
def read():
    ph, span, nw = (C.c_double * 76)(), C.c_double(), C.c_uint64()
    _lib.check(L.c4_debug_tower_phases(ph, 76, C.byref(span), C.byref(nw), 1))
    return list(ph), span.value, nw.value
for _ in range(20):
    net.tower(x)
read()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(200):
    net.tower(x)
b.record(); torch.cuda.synchronize()
ph, _, nw = read()
print(f"tower {ch} ch, {blocks} blocks, {n} boards: {a.elapsed_time(b) / 200 * 1e3:.1f} us per launch (eager, back to back), {nw} workgroups")
if ch == 64:
    print(f"  mean workgroup (its first wavefront), us: entry->input staged {ph[0]:.2f}  conv0 {ph[1]:.2f}")
    tot = ph[0] + ph[1]
    for l in range(1, 2 * blocks + 1):
        k, e, h = ph[2 + 3 * (l - 1): 2 + 3 * l]
        tot += k + e + h
        print(f"    layer {l:2d}: k-loop {k:.2f}  epilogue {e:.2f}  pair hand-over {h:.2f}  | {k + e + h:.2f}")
    print(f"  sum {tot:.2f}")
    print("  lifetime by wavefront number, us (mean over workgroups): " + "  ".join(f"w{w} {ph[61 + w]:.1f}" for w in range(8)))
else:
    names = ["entry->input staged", "conv0"] + [f"layer {i}" for i in range(1, 2 * blocks + 1)]
    print("  mean workgroup, us: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(names, ph)) + f"  | sum {sum(ph[:len(names)]):.2f}")
if ph[70] > 0:
    print(f"  shader clock over a workgroup's life (its first wavefront; s_memtime / s_memrealtime): {ph[69] / ph[70] * 0.1:.2f} GHz")
spans = []
for _ in range(20):
    torch.cuda.synchronize(); net.tower(x); spans.append(read()[1])
spans.sort()
print(f"  in-kernel span of one launch: median {spans[10]:.2f} us, min {spans[0]:.2f}")
