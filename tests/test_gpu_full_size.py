"""GPU tests at BASELINE.json's full sizes: size-independent properties of the domain (every game is
checked structurally; the same bytes whatever the placement) AND, since round 6, every one of config 2's
4 096 games against the oracle bit for bit (the C oracle in the reference's thread topology plays the
6 M simulations in a few seconds), plus the reference's default search width n_mcts_iterations = 1400
(src/c4a0/main.py:41)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _run(n_games, n_slots, n_iter, ids=None):
    from c4a0_amd.session import DeviceSession
    from tests.helpers import hash_eval_torch

    ids = list(range(n_games)) if ids is None else ids
    s = DeviceSession(n_slots, n_iter, 6.6, 0.01, planes_dtype=torch.bfloat16)
    s.set_games([(g, 0, 0) for g in ids])
    s.run(hash_eval_torch, poll_every=64)
    recs, counts, ctr = s.drain_samples(), s.sample_counts(), s.counters()
    s.close()
    return recs, counts, ctr


def _check_structure(recs, counts, ids):
    from c4a0_amd.results import terminal_state

    assert counts.min() >= 8 and counts.max() <= 43            # >= 7 moves to win + terminal sample
    offs = np.concatenate([[0], np.cumsum(counts.astype(np.int64))]).astype(np.int64)
    meta = recs["meta"]
    assert np.array_equal(recs["game_id"], np.repeat(np.array(ids, dtype=np.uint64), counts))
    # policies: probability vectors that vanish on full columns; uniform 1/7 on the terminal sample
    pol = recs["policy"]
    assert np.all(np.abs(pol.sum(1, dtype=np.float32) - 1.0) < 1e-5) and pol.min() >= 0.0
    full = ((recs["mask"][:, None] >> (np.uint64(35) + np.arange(7, dtype=np.uint64))) & np.uint64(1)).astype(bool)
    is_term = (meta >> 16) == 1
    assert np.all(pol[full & ~is_term[:, None]] == 0.0)
    assert np.all(pol[is_term] == np.float32(1.0) / np.float32(7.0))
    popc = np.array([bin(int(m)).count("1") for m in recs["mask"]])
    for gi in range(len(ids)):
        a, b = offs[gi], offs[gi + 1]
        m = b - a - 1
        assert np.array_equal(meta[a:b] & 0xFFFF, np.arange(m + 1))          # sample indices 0..M
        assert is_term[b - 1] and not is_term[a:b - 1].any()                  # exactly one terminal sample, last
        assert recs["mask"][a] == 0 and np.array_equal(popc[a:b], np.arange(m + 1))   # one piece per move from the empty board
        t = terminal_state(int(recs["mask"][b - 1]), int(recs["value"][b - 1]))
        assert t in (2, 3)                                                     # OpponentWin or Draw (c4r.rs:228-238)
        qp, qn = recs["q_penalty"][b - 1], recs["q_no_penalty"][b - 1]
        assert qn == (np.float32(-1.0) if t == 2 else np.float32(0.0))
        if t == 2:
            assert qp == np.float32(-1.0) + np.float32(0.01) * np.float32(popc[b - 1])   # c4r.rs:253-263
        sign = np.where((m - np.arange(m + 1)) % 2 == 0, np.float32(1), np.float32(-1))   # mcts.rs:279-298
        assert np.array_equal(recs["q_penalty"][a:b], sign * qp) and np.array_equal(recs["q_no_penalty"][a:b], sign * qn)
        # consecutive positions differ by one legal move, seen from the other side (c4r.rs:58-72)
        mk, vl = recs["mask"][a:b].astype(object), recs["value"][a:b].astype(object)
        for i in range(m):
            new = int(mk[i + 1]) ^ int(mk[i])
            assert bin(new).count("1") == 1 and int(vl[i + 1]) == (~(int(vl[i]) | new)) & int(mk[i + 1])


def test_config2_size_properties_placement_independence_and_oracle_subset():
    """BASELINE config 2 shape: 4 096 concurrent games, n_mcts_iterations = 100."""
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game, samples_by_game

    n = 4096
    ids = list(range(n))
    recs, counts, ctr = _run(n, 4096, 100)
    assert ctr["games_done"] == n and ctr["error"] == 0 and ctr["samples"] == len(recs) == counts.sum()
    _check_structure(recs, counts, ids)
    # idempotence / placement independence: 1 024 slots (every slot replays 4 games) gives the same bytes
    recs2, counts2, ctr2 = _run(n, 1024, 100)
    assert recs2.tobytes() == recs.tobytes() and np.array_equal(counts, counts2)
    assert {k: ctr[k] for k in ("sims", "select_levels", "backup_nodes", "expansions", "moves")} == \
           {k: ctr2[k] for k in ("sims", "select_levels", "backup_nodes", "expansions", "moves")}
    # EVERY game replayed by the oracle, bit for bit (round 6: all 4 096; 128 in round 5's run, 48 before -- the C oracle in the
    # reference's thread topology plays config 2's 6 M simulations in a few seconds on the GPU box's host cores)
    sub = ids
    want, _ = O.self_play([(g, 0, 0) for g in sub], 4096, 100, 6.6, 0.01, "hash", n_threads=max(2, min(16, os.cpu_count() or 2)), topology="async")
    got = samples_by_game(recs)
    assert got == oracle_samples_by_game(want)
    from tests.helpers import evidence
    evidence(f"config 2 (4 096 games, n = 100, hash evaluator): {len(recs)} samples structurally checked, 2 placements byte-identical, ALL {len(sub)} games == oracle bit for bit")
    # every game needs at least 7 moves x (n - retained) sims
    assert ctr["sims"] / n > 300 and 8 <= len(recs) / n <= 43


def test_reference_default_search_width_1400():
    """n_mcts_iterations = 1400 (the reference's CLI default): worst-case arena sizing and deep
    trees; two games against the oracle."""
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game, samples_by_game

    ids = [11, 12]
    recs, counts, ctr = _run(2, 2, 1400, ids)
    want, ost = O.self_play([(g, 0, 0) for g in ids], 64, 1400, 6.6, 0.01, "hash")
    assert samples_by_game(recs) == oracle_samples_by_game(want)
    assert ctr["backup_nodes"] == ost["backup_nodes"] and ctr["expansions"] == ost["expansions"]
