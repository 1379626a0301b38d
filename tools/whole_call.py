#!/usr/bin/env python3
"""Whole `play_games(evaluator=)` calls at a BASELINE shape, with the call's wall time split into phases (stats["phases"]):
    python tools/whole_call.py N_GAMES [resident,resident,...] [--n-mcts 100] [--blocks 4] [--channels 32] [--reps 2] [--pickle]"""
import argparse
import json
import os
import pickle
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n_games", type=int)
    ap.add_argument("resident", nargs="?", default="0")
    ap.add_argument("--n-mcts", type=int, default=100)
    ap.add_argument("--blocks", type=int, default=4)
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--pickle", action="store_true")
    ap.add_argument("--sessions", type=int, default=0)
    ap.add_argument("--host-loop", default=None, choices=["native", "python"], help="default: play_games' own choice (the library's loop for an InferenceNet)")
    ap.add_argument("--check-python-loop", action="store_true", help="play the job once more with host_loop='python' and compare the records")
    a = ap.parse_args()
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    dev = torch.device("cuda:0")
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(a.blocks, a.channels, 4, 2)), dev, dtype=torch.bfloat16)
    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(a.n_games)]
    c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, evaluator=net)     # untimed: code objects, LDS opt-ins
    for res_games in [int(x) for x in a.resident.split(",")]:
        for rep in range(a.reps):
            st = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = c4a0_amd.play_games(reqs, 2000, a.n_mcts, 6.6, 0.01, evaluator=net, stats=st, resident_games=res_games or None,
                                      concurrent_sessions=a.sessions or None, host_loop=a.host_loop)
            dt = time.perf_counter() - t0
            out = {"n_games": a.n_games, "resident_games": res_games, "rep": rep, "host_loop": st.get("host_loop"), "sessions": st.get("concurrent_sessions"),
                   "seconds": round(dt, 4), "games_per_s": round(a.n_games / dt, 1),
                   "steps": st["steps"], "sims": st["sims"], "n_slots": st["n_slots"],
                   "phases": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in st["phases"].items()}}
            if a.pickle:
                t0 = time.perf_counter()
                blob = pickle.dumps(res)
                out["pickle_dumps_s"] = round(time.perf_counter() - t0, 4)
                t0 = time.perf_counter()
                pickle.loads(blob)
                out["pickle_loads_s"] = round(time.perf_counter() - t0, 4)
                out["pickle_bytes"] = len(blob)
            if a.check_python_loop and rep == 0:
                st2 = {}
                t0 = time.perf_counter()
                res2 = c4a0_amd.play_games(reqs, 2000, a.n_mcts, 6.6, 0.01, evaluator=net, stats=st2, resident_games=res_games or None,
                                           concurrent_sessions=a.sessions or None, host_loop="python")
                out["python_loop_seconds"] = round(time.perf_counter() - t0, 4)
                out["python_loop_records_identical"] = bool(res2.to_records()[0].tobytes() == res.to_records()[0].tobytes())
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
