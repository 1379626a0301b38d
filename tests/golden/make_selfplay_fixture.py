"""Generates tests/golden/selfplay_hash.json: whole self-play games produced by the CPU oracle
(oracle/c4_oracle.c, itself pinned by the reference's known-answer tests) under the integer-hash
evaluator, for fixed game ids.  Committed so that both the oracle (CPU suite) and the HIP path
(GPU suite) are checked against the same recorded vectors, not only against each other.

    python tests/golden/make_selfplay_fixture.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import c4oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "selfplay_hash.json")
CASES = [  # (game ids, n_mcts_iterations, c_exploration, c_ply_penalty)
    ([0, 1, 42, 43, 1764, (1 << 64) - 1], 10, 6.6, 0.01),
    ([7, 8], 100, 6.6, 0.01),
    ([3], 37, 1.4, 0.05),
]


def bits(x):
    return int(np.float32(x).view(np.uint32))


def main():
    out = []
    for ids, n, c_expl, c_ply in CASES:
        res, st = O.self_play([(g, 0, 0) for g in ids], 64, n, c_expl, c_ply, "hash")
        games = {str(g): [[s.mask, s.value, [bits(p) for p in s.policy], bits(s.q_penalty), bits(s.q_no_penalty)] for s in res[g]]
                 for g in ids}
        out.append({"ids": ids, "n_mcts_iterations": n, "c_exploration": c_expl, "c_ply_penalty": c_ply,
                    "evaluator": "hash", "games": games,
                    "counters": {k: st[k] for k in ("sims", "sims_terminal_root", "select_levels", "select_levels_discarded",
                                                    "backup_nodes", "expansions", "moves", "n_samples")}})
    json.dump(out, open(OUT, "w"), separators=(",", ":"))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
