"""Committed golden vectors (tests/golden/selfplay_hash.json, made by make_selfplay_fixture.py):
the oracle must keep reproducing them (CPU), and the HIP path must reproduce them (GPU)."""
import json
import os

import numpy as np
import pytest

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "selfplay_hash.json")))


def _bits(x):
    return int(np.float32(x).view(np.uint32))


@pytest.mark.parametrize("case", range(len(FIX)))
def test_oracle_reproduces_golden_games(case):
    from oracle import c4oracle as O

    c = FIX[case]
    res, st = O.self_play([(g, 0, 0) for g in c["ids"]], 64, c["n_mcts_iterations"], c["c_exploration"], c["c_ply_penalty"], "hash")
    for g in c["ids"]:
        got = [[s.mask, s.value, [_bits(p) for p in s.policy], _bits(s.q_penalty), _bits(s.q_no_penalty)] for s in res[g]]
        assert got == c["games"][str(g)], g
    assert {k: st[k] for k in c["counters"]} == c["counters"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(FIX)))
def test_hip_reproduces_golden_games(case):
    torch = pytest.importorskip("torch")
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch

    c = FIX[case]
    s = DeviceSession(4, c["n_mcts_iterations"], c["c_exploration"], c["c_ply_penalty"])
    s.set_games([(g, 0, 0) for g in c["ids"]])
    s.run(hash_eval_torch)
    recs = s.drain_samples()
    ctr = s.counters()
    s.close()
    for g in c["ids"]:
        rows = recs[recs["game_id"] == np.uint64(g)]
        rows = rows[np.argsort(rows["meta"] & 0xFFFF)]
        got = [[int(r["mask"]), int(r["value"]), [int(x) for x in r["policy"].view(np.uint32)],
                int(r["q_penalty"].view(np.uint32)), int(r["q_no_penalty"].view(np.uint32))] for r in rows]
        assert got == c["games"][str(g)], g
    k = c["counters"]
    assert ctr["sims"] + ctr["ref_skipped_sims"] == k["sims"] and ctr["ref_skipped_sims"] == k["sims_terminal_root"]
    assert ctr["select_levels"] == k["select_levels"] - k["select_levels_discarded"]
    assert (ctr["backup_nodes"], ctr["expansions"], ctr["moves"], ctr["samples"]) == \
           (k["backup_nodes"], k["expansions"], k["moves"], k["n_samples"])
