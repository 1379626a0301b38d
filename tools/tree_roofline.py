#!/usr/bin/env python3
"""Tree-path roofline sweep: the fused step kernel alone (no network), G games per launch.

SURVEY 8(d) "honest expectation": at BASELINE config 2's G = 4096 one launch moves ~3 MB, far
below what hides HBM latency, so the >= 70 % HBM target is only meaningful at large G.  This
tool grows real trees with the uniform evaluator (logits and q constant, evaluated once: the
reference's UniformEvalPos, self_play.rs:391-403) at n_mcts_iterations = 100 and times K
launches per G with HIP events and the in-kernel device clock.

    python tools/tree_roofline.py [--games 4096,16384,65536,262144] [--steps 300] > gpurun_out/tree_roofline.json
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import HBM_PEAK_GBPS, algorithmic_bytes  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", default="4096,16384,65536,131072")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--preroll", type=int, default=2500)
    ap.add_argument("--n-mcts", type=int, default=100)
    ap.add_argument("--blocks-per-slot", type=int, default=0, help="0 = worst case 43*n+8")
    args = ap.parse_args()
    from c4a0_amd.session import DeviceSession

    dev = torch.device("cuda:0")
    out = []
    for g in [int(x) for x in args.games.split(",")]:
        s = DeviceSession(g, args.n_mcts, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16, blocks_per_slot=args.blocks_per_slot)
        n_games = g * 8
        ids = np.zeros((n_games, 3), dtype=np.uint64)
        ids[:, 0] = np.arange(n_games, dtype=np.uint64)
        s.set_games(ids)
        s.bind()
        s.start()
        s.logprobs.fill_(1.0 / 7.0)
        s.q.zero_()
        for _ in range(args.preroll):
            s.step()
        c0 = s.counters()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in ev:
            a.record()
            s.step()
            b.record()
        torch.cuda.synchronize()
        c1 = s.counters()
        assert c1["error"] == 0, c1
        d = {k: c1[k] - c0[k] for k in c1 if k not in ("error", "error_slot")}
        ab = algorithmic_bytes(d, 2)
        ev_us = sum(a.elapsed_time(b) for a, b in ev) * 1e3 / args.steps
        dev_us = d["step_kernel_ns"] / 1e3 / max(1, d["step_launches"])
        per_launch = ab["total"] / args.steps
        out.append({"games_per_launch": g, "active_sims_per_launch": d["sims"] / args.steps,
                    "algorithmic_bytes_per_launch": per_launch, "bytes_per_sim": ab["total"] / max(1, d["sims"]),
                    "select_backup_bytes_per_sim": ab["select_backup"] / max(1, d["sims"]),
                    "S": d["select_levels"] / max(1, d["sims"]), "K": d["backup_nodes"] / max(1, d["sims"]), "E": d["expansions"] / max(1, d["sims"]),
                    "event_us": ev_us, "device_clock_us": dev_us,
                    "GBps_events": per_launch / ev_us / 1e3, "GBps_device_clock": per_launch / dev_us / 1e3,
                    "frac_events": per_launch / ev_us / 1e3 / HBM_PEAK_GBPS, "frac_device_clock": per_launch / dev_us / 1e3 / HBM_PEAK_GBPS,
                    "sims_per_s_events": d["sims"] / args.steps / ev_us * 1e6})
        print(json.dumps(out[-1]), file=sys.stderr)
        s.close()
        del s
        torch.cuda.empty_cache()
    print(json.dumps({"kernel": "c4_step_kernel", "evaluator": "uniform (constant)", "n_mcts_iterations": args.n_mcts, "peak_GBps": HBM_PEAK_GBPS, "sweep": out}))


if __name__ == "__main__":
    main()
