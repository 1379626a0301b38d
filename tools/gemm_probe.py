"""The heads' hidden-layer GEMMs: the hand-written MFMA kernel (c4_linear_bf16, every tile configuration)
against PyTorch's hipBLASLt, (a) one GEMM alone, back to back, and (b) the chain of a whole head pass
(N = 2F, F, F) replayed from HIP graphs on TWO streams at once -- what a session sees while the other
session shares the chip.   python tools/gemm_probe.py [M] [out.json]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()


def hip_linear(x, w, b32, cfg, y=None):
    m, n, k = x.shape[0], w.shape[0], w.shape[1]
    y = y if y is not None else torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    _lib.check(L.c4_linear_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b32.data_ptr()), C.c_void_p(y.data_ptr()),
                                m, n, k, x.stride(0), y.stride(0), 1, cfg, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return y


def timeit(fn, n=200):
    """fn captured in a HIP graph of 20 calls (no host launch cost in the figure), replayed n/20 times."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n // 20):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n // 20 * 20) * 1e3


def chain_pair(make_chain, n_rep=10, per_graph=10):
    """Two streams, each replaying a graph of `per_graph` head passes; us per pass per stream-pair."""
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    graphs, keep = [], []
    for st in streams:
        fn = make_chain()
        keep.append(fn)        # the operands live in fn's closure: they must outlive the graph
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(3):
                fn()
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(per_graph):
                fn()
        graphs.append(g)
    torch.cuda.synchronize()
    for st, g in zip(streams, graphs):
        with torch.cuda.stream(st):
            g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    for _ in range(n_rep):
        for st, g in zip(streams, graphs):
            with torch.cuda.stream(st):
                g.replay()
    for st in streams:
        torch.cuda.current_stream().wait_stream(st)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (n_rep * per_graph) * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    F = int(os.environ.get("PROBE_F", "1344"))   # 1344 = 42 x 32 channels, 2688 = 42 x 64
    out = {"M": M, "alone_us": {}, "pair_chain_us": {}}
    torch.manual_seed(0)
    for N in (2 * F, F):
        x = torch.randn(M, F, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, F, device=dev) / F ** 0.5).to(torch.bfloat16)
        b = torch.randn(N, device=dev).to(torch.bfloat16)
        b32 = b.float()
        y = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        row = {"hipblaslt": timeit(lambda: torch._addmm_activation(b, x, w.t(), use_gelu=False))}
        for cfg in ([int(c) for c in os.environ["PROBE_CFGS"].split(",")] if "PROBE_CFGS" in os.environ else
                    range(1, 17) if "PROBE_ALL" in os.environ else (4, 9, 10, 11, 17, 18, 19)):
            row[f"hip{cfg}"] = timeit(lambda: hip_linear(x, w, b32, cfg, y))
        fl = 2 * M * F * N
        out["alone_us"][f"N{N}"] = {k: round(v, 2) for k, v in row.items()}
        print(f"M{M} N{N} alone:", " ".join(f"{k}={v:.1f}us({fl / v / 1e6:.0f}TF)" for k, v in row.items()), flush=True)

    def chain_factory(kind, cfgs=None):
        def make():
            x = torch.randn(M, F, device=dev).to(torch.bfloat16)
            w1 = (torch.randn(2 * F, F, device=dev) / F ** 0.5).to(torch.bfloat16)
            w2 = (torch.randn(F, F, device=dev) / F ** 0.5).to(torch.bfloat16)
            w3 = (torch.randn(F, F, device=dev) / F ** 0.5).to(torch.bfloat16)
            b1, b2 = torch.randn(2 * F, device=dev).to(torch.bfloat16), torch.randn(F, device=dev).to(torch.bfloat16)
            b1f, b2f = b1.float(), b2.float()
            h1 = torch.empty((M, 2 * F), dtype=torch.bfloat16, device=dev)
            h2 = torch.empty((M, F), dtype=torch.bfloat16, device=dev)
            h3 = torch.empty((M, F), dtype=torch.bfloat16, device=dev)
            if kind == "hipblaslt":
                def fn():
                    h = torch._addmm_activation(b1, x, w1.t(), use_gelu=False)
                    p = torch._addmm_activation(b2, h[:, :F], w2.t(), use_gelu=False)
                    return torch._addmm_activation(b2, p, w3.t(), use_gelu=False)
            else:
                c1, c2 = cfgs
                def fn():
                    hip_linear(x, w1, b1f, c1, h1)
                    hip_linear(h1[:, :F], w2, b2f, c2, h2)
                    return hip_linear(h2, w3, b2f, c2, h3)
            return fn
        return make

    res = {}
    only = os.environ.get("PROBE_ONLY")
    if only == "alone":
        return
    combos = [(1, 1), (2, 2), (3, 3), (1, 4), (4, 4), (5, 5), (6, 6), (6, 4), (6, 10), (10, 10), (7, 7), (8, 8), (9, 9), (2, 9)]
    if only != "hip":
        res["hipblaslt"] = chain_pair(chain_factory("hipblaslt"))
        print("pair hipblaslt", res["hipblaslt"], flush=True)
    for c1, c2 in ([] if only == "hipblaslt" else combos):
        res[f"hip{c1}_{c2}"] = chain_pair(chain_factory("hip", (c1, c2)))
        print(f"pair hip{c1}_{c2}", res[f"hip{c1}_{c2}"], flush=True)
    out["pair_chain_us"] = {k: round(v, 2) for k, v in res.items()}
    print(f"M{M} two streams, one head pass (N=2F,F,F) each:", " ".join(f"{k}={v:.1f}" for k, v in res.items()), flush=True)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(json.dumps(out) + "\n")


if __name__ == "__main__":
    main()
