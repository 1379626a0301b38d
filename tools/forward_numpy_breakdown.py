#!/usr/bin/env python3
"""Where a `forward_numpy` call goes (the callback of the reference's callers): host time of each statement and the
whole call, for pageable and for pinned input, by batch size.   python tools/forward_numpy_breakdown.py [rows ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

dev = torch.device("cuda:0")
torch.manual_seed(1337)
net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), dev, dtype=torch.bfloat16)
for rows in [int(a) for a in sys.argv[1:]] or [1700, 900, 256]:
    x = (np.random.default_rng(1).random((rows, 2, 6, 7)) < 0.3).astype(np.float32)
    xp = torch.from_numpy(x).pin_memory().numpy()
    for name, arr in (("pageable", x), ("pinned", xp)):
        for _ in range(20):
            net.forward_numpy(arr)
        t0 = time.perf_counter()
        for _ in range(300):
            net.forward_numpy(arr)
        whole = (time.perf_counter() - t0) / 300
        # the statements of the call, one by one (each followed by a stream synchronisation: upper bounds)
        st = net._np
        b, bucket = rows, -(-rows // net._NP_BUCKET) * net._NP_BUCKET
        g = st["graphs"][bucket]
        parts = {}
        def timed(label, fn, n=200):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
                st["stream"].synchronize()
            parts[label] = (time.perf_counter() - t) / n * 1e6
        with torch.cuda.stream(st["stream"]):
            timed("copy to device (pageable input only)", lambda: st["in"][:b].copy_(torch.from_numpy(arr), non_blocking=True))
            timed("is_pinned", lambda: torch.from_numpy(arr).is_pinned())
            timed("graph", lambda: g.replay())
            timed("sync only", lambda: None)
            import ctypes as C
            from c4a0_amd._lib import check
            def eager():
                check(net._L.c4_planes_from_f32(C.c_void_p(st["slot"].data_ptr()), None, 0, C.c_void_p(st["planes"].data_ptr()), bucket,
                                                C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
                net.forward(st["planes"][:bucket], out_logprobs=st["h_lp"][:bucket], out_q=st["h_q"][:bucket])
            saved, net.latency_mode = net.latency_mode, True
            timed("the same launches without a graph", eager)
            net.latency_mode = saved
        t = time.perf_counter()
        for _ in range(2000):
            lp, q = st["h_lp"].numpy()[:b], st["h_q"].numpy()[:b]
            out = lp.copy(), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])
        parts["numpy out"] = (time.perf_counter() - t) / 2000 * 1e6
        print(f"{rows:5d} rows {name:8s}: whole call {whole * 1e6:6.1f} us;  " + "  ".join(f"{k} {v:.1f}" for k, v in parts.items()), flush=True)
