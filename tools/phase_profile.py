#!/usr/bin/env python3
"""Where a step-kernel launch spends its time (diagnostic build with in-kernel phase stamps).

    C4A0_HIP_LIB=libc4a0_hip_diag.so python tools/phase_profile.py [--games 4096] [--with-nn]

Prints, for the last launch, the median and max over wavefronts of the time between phase
boundaries (100 MHz device clock).  Read the SHARES, not the length: the stamps' waits forbid
overlaps the product kernel has."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

os.environ.setdefault("C4A0_HIP_LIB", "libc4a0_hip_diag.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NAMES = ["start->state loaded", "expand", "backup", "fence", "gate/move", "select", "2nd trip + encode + state store", "counters"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--preroll", type=int, default=2500)
    ap.add_argument("--n", type=int, default=100, help="n_mcts_iterations (1400 = the reference's default job: deep trees)")
    ap.add_argument("--blocks", type=int, default=4, help="residual blocks of the network (--with-nn)")
    ap.add_argument("--with-nn", action="store_true", help="run the real evaluator between launches (cold caches)")
    args = ap.parse_args()
    from c4a0_amd import _lib
    from c4a0_amd.session import DeviceSession

    dev = torch.device("cuda:0")
    s = DeviceSession(args.games, args.n, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games([(i, 0, 0) for i in range(args.games * 8)])
    s.bind()
    s.start()
    ev = None
    if args.with_nn:
        from c4a0_amd.nn import ConnectFourNet, GraphedEvaluator, InferenceNet, ModelConfig
        torch.manual_seed(1337)
        net = InferenceNet(ConnectFourNet(ModelConfig(args.blocks, 32, 4, 2)), dev)
        ev = GraphedEvaluator(net, s.planes, s.logprobs, s.q)
    else:
        s.logprobs.fill_(1.0 / 7.0)
    acc, acc_full = [], []
    for i in range(args.preroll + 50):
        if ev is not None:
            s.evaluate(ev)
        s.step()
        if i >= args.preroll:
            n = C.c_uint64()
            _lib.check(s.L.c4_session_debug_phase_stamps(s._h, None, 0, C.byref(n)))
            buf = np.zeros(n.value, dtype=np.uint64)
            _lib.check(s.L.c4_session_debug_phase_stamps(s._h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
            acc.append(buf.reshape(-1, 16)[:, :9].astype(np.int64))
            acc_full.append(buf.reshape(-1, 16).astype(np.int64))
    st = np.stack(acc)  # [launch, wave, 9]
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    print(f"games={args.games} n={args.n} with_nn={args.with_nn}; times in us (10 ns ticks), over {st.shape[0]} launches x {st.shape[1]} wavefronts")
    print(f"  wave start spread: median {np.median(st[:, :, 0] - t0) / 100:.2f}  max {np.max(st[:, :, 0] - t0) / 100:.2f}")
    for k, name in enumerate(NAMES):
        dt = (st[:, :, k + 1] - st[:, :, k]) / 100.0
        print(f"  {name:24s} median {np.median(dt):6.2f}  p90 {np.percentile(dt, 90):6.2f}  max {dt.max():6.2f}")
    full = np.stack(acc_full)
    mv = full[:, :, 9] > 0   # wavefronts that executed the move phase in that launch
    if mv.any():
        seg = [("gate -> root policy", 4, 9), ("temperature", 9, 10), ("chacha12 + sample", 10, 11), ("record/re-root/finish", 11, 12), ("refill -> select start", 12, 5)]
        print(f"  move phase, {int(mv.sum())} (launch, wavefront) pairs with a mover:")
        for name, a, b in seg:
            dt = (full[:, :, b] - full[:, :, a])[mv] / 100.0
            print(f"    {name:24s} median {np.median(dt):6.2f}  max {dt.max():6.2f}")
    t2 = full[:, :, 13] > 0   # wavefronts in which some game ran a second simulation (terminal leaf)
    if t2.any():
        dt = (full[:, :, 13] - full[:, :, 6])[t2] / 100.0
        print(f"  second trip (terminal leaf): {100.0 * t2.mean():.0f} % of wavefronts, median {np.median(dt):.2f}  p90 {np.percentile(dt, 90):.2f}  max {dt.max():.2f}")
    tot = (st[:, :, 8].max(axis=1) - st[:, :, 0].min(axis=1)) / 100.0
    print(f"  launch (first start -> last end): mean {tot.mean():.2f}")
    s.close()


if __name__ == "__main__":
    main()
