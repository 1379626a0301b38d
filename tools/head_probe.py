#!/usr/bin/env python3
"""Times the heads' output kernel (c4_head_out_bf16) alone on hidden activations of BASELINE config 2's shape."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd import _lib
L = _lib.lib()
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1344
both = torch.randn(n, 2 * F, device=dev).bfloat16()
hp, hv = both[:, :F], both[:, F:]
wp = torch.randn(7, F, device=dev).bfloat16(); wv = torch.randn(2, F, device=dev).bfloat16()
bp = torch.randn(7, device=dev); bv = torch.randn(2, device=dev)
lp = torch.empty(n, 7, device=dev); q = torch.empty(n, 2, device=dev)
big = torch.empty(64 << 20, device=dev)   # 256 MB: push the activations out of L2 between calls
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    _lib.check(L.c4_head_out_bf16(C.c_void_p(hp.data_ptr()), C.c_void_p(hv.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(wv.data_ptr()),
                                  C.c_void_p(bp.data_ptr()), C.c_void_p(bv.data_ptr()), n, F, hp.stride(0), hv.stride(0),
                                  C.c_void_p(lp.data_ptr()), C.c_void_p(q.data_ptr()), st))
for _ in range(5): call()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(100): call()
b.record(); torch.cuda.synchronize()
hot = a.elapsed_time(b) / 100 * 1e3
tot = 0.0
for _ in range(20):
    big.zero_()
    a.record(); call(); b.record(); torch.cuda.synchronize()
    tot += a.elapsed_time(b) * 1e3
print(f"head out n={n} F={F}: back-to-back {hot:.1f} us, after a 256 MB sweep {tot / 20:.1f} us per call")
