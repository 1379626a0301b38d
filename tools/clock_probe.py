#!/usr/bin/env python3
"""Which shader clock do the head GEMMs actually get?  Diagnostic build only (python c4a0_amd/csrc/build.py --diag):
c4_head_gemm_kernel stamps s_memtime / s_memrealtime around its main loop and c4_debug_gemm_clock() reports the mean.

  (a) the wide layer alone, back to back (224 workgroups at 2 048 rows)
  (b) a narrow layer alone (112 workgroups)
  (c) the whole evaluator pass of ONE session, back to back
  (d) two sessions' evaluator passes on two streams (what the bench runs, minus the step kernel)
Prints the clock (GHz), the main-loop time per workgroup (us) and the cycles per 64-deep k-tile."""
import ctypes as C
import os
import sys

os.environ.setdefault("C4A0_HIP_LIB", "libc4a0_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from c4a0_amd import _lib  # noqa: E402
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
L.c4_debug_gemm_clock.restype = C.c_int
L.c4_debug_gemm_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
L.c4_debug_gemm_phases.restype = C.c_int
L.c4_debug_gemm_phases.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int]
PHASES = ["entry->prologue issued", "->first k-tile landed", "->main loop done", "->tail DMA drained", "->bias there", "->stores issued", "->stores acknowledged"]


def phases(reset=True):
    ph, span = (C.c_double * 7)(), C.c_double()
    _lib.check(L.c4_debug_gemm_phases(ph, C.byref(span), 1 if reset else 0))
    return list(ph), span.value


def read(reset=True):
    g, u, n = C.c_double(), C.c_double(), C.c_uint64()
    _lib.check(L.c4_debug_gemm_clock(C.byref(g), C.byref(u), C.byref(n), 1 if reset else 0))
    return g.value, u.value, n.value


def linear(x, w, b32, y, cfg):
    m, n, k = x.shape[0], w.shape[0], w.shape[1]
    _lib.check(L.c4_linear_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b32.data_ptr()), C.c_void_p(y.data_ptr()),
                                m, n, k, x.stride(0), y.stride(0), 1, cfg, C.c_void_p(torch.cuda.current_stream().cuda_stream)))


REPS = 300


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    cfgs = [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["11"])]
    F = 1344
    torch.manual_seed(0)
    x = torch.randn(M, F, device=dev).to(torch.bfloat16)
    for cfg in cfgs:
        for N in (2 * F, F):
            w = (torch.randn(N, F, device=dev) / F ** 0.5).to(torch.bfloat16)
            b32 = torch.randn(N, device=dev)
            y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
            for _ in range(100):     # warm: clocks settle after ~ms of load
                linear(x, w, b32, y, cfg)
            read()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(REPS):
                linear(x, w, b32, y, cfg)
            b.record()
            torch.cuda.synchronize()
            ghz, us, n = read(reset=False)
            ph, _ = phases()
            print(f"cfg {cfg} M {M} N {N} alone: {a.elapsed_time(b) / REPS * 1e3:.1f} us per launch; main loop {us:.2f} us at {ghz:.3f} GHz "
                  f"= {us * ghz * 1e3 / 21:.0f} cycles per k-tile ({n} workgroups)", flush=True)
            print("    mean workgroup, us: " + "  ".join(f"{nm} {v:.2f}" for nm, v in zip(PHASES, ph)), flush=True)
            spans = []
            for _ in range(20):          # one launch at a time: first workgroup's entry -> last workgroup's last store acknowledged
                torch.cuda.synchronize()
                linear(x, w, b32, y, cfg)
                spans.append(phases()[1])
            spans.sort()
            print(f"    in-kernel span of one launch (first entry -> last exit): median {spans[10]:.2f} us, min {spans[0]:.2f}", flush=True)
    if len(sys.argv) > 3 and sys.argv[3] == "gemm-only":
        return
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), dev, dtype=torch.bfloat16)
    planes = (torch.rand(M, 2, 6, 7, device=dev) > 0.7).to(torch.bfloat16)
    lp, q = torch.empty(M, 7, device=dev), torch.empty(M, 2, device=dev)
    for _ in range(100):
        net(planes, out_logprobs=lp, out_q=q)
    read()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(1000):
        net(planes, out_logprobs=lp, out_q=q)
    b.record()
    torch.cuda.synchronize()
    ghz, us, n = read()
    print(f"one session's evaluator pass, back to back: {a.elapsed_time(b):.1f} us per pass; GEMM main loops {us:.2f} us at {ghz:.3f} GHz", flush=True)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = [(planes.clone(), torch.empty(M, 7, device=dev), torch.empty(M, 2, device=dev)) for _ in streams]
    graphs = []
    for st, (pl, l_, q_) in zip(streams, bufs):
        with torch.cuda.stream(st):
            for _ in range(3):
                net(pl, out_logprobs=l_, out_q=q_)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(16):
                net(pl, out_logprobs=l_, out_q=q_)
        graphs.append(g)
    for _ in range(5):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    torch.cuda.synchronize()
    read()
    a.record()
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for _ in range(60):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    b.record()
    torch.cuda.synchronize()
    ghz, us, n = read()
    print(f"two sessions' evaluator passes on two streams: {a.elapsed_time(b) / 960 * 1e3:.1f} us per pair of passes; GEMM main loops {us:.2f} us at {ghz:.3f} GHz", flush=True)


if __name__ == "__main__":
    main()
