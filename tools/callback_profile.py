#!/usr/bin/env python3
"""cProfile of a numpy-callback job (host side): where play_games' own Python time goes per step."""
import cProfile, os, pstats, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import c4a0_amd
from tests.helpers import hash_eval_np

games, n_mcts = int(sys.argv[1]) if len(sys.argv) > 1 else 1700, int(sys.argv[2]) if len(sys.argv) > 2 else 200
reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(games)]
def cb(m, x):
    n = x.shape[0]
    return np.zeros((n, 7), np.float32) - 1.9459101, np.zeros(n, np.float32), np.zeros(n, np.float32)
c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, cb)
pr = cProfile.Profile()
st = {}
pr.enable()
c4a0_amd.play_games(reqs, 2000, n_mcts, 6.6, 0.01, cb, stats=st)
pr.disable()
print("steps", st["steps"])
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
