#!/usr/bin/env python3
"""T3 without subsets at any BASELINE shape: one generation of N games with the real bf16 network; the run logs what the
evaluator answered for every row of every step, the distinct positions become the oracle's evaluator (c4o_eval_table) and EVERY game's
samples must equal the oracle's bit for bit (tests/test_gpu_baseline_configs.py runs this for config 2 inside the GPU suite; the larger
shapes take minutes and are run from here -> profiles/r06_full_t3.txt).

    python tools/full_t3.py BLOCKS CHANNELS N_GAMES N_MCTS [ALPHA EPS]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    blocks, channels, n, n_iter = (int(x) for x in sys.argv[1:5])
    dirichlet = (float(sys.argv[5]), float(sys.argv[6])) if len(sys.argv) > 6 else None
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game, samples_by_game
    from tests.test_gpu_baseline_configs import _net, _run_logging_every_row
    from tests.test_gpu_full_size import _check_structure

    ids = list(range(n))
    t0 = time.time()
    recs, counts, ctr, table, (n_rows, n_dup) = _run_logging_every_row(_net(blocks, channels), ids, n, n_iter, dirichlet=dirichlet)
    t1 = time.time()
    assert ctr["games_done"] == n and ctr["error"] == 0
    _check_structure(recs, counts, ids)
    want, _ = O.self_play([(g, 0, 0) for g in ids], 4096, n_iter, 6.6, 0.01, ("table",) + table, n_threads=max(2, min(16, os.cpu_count() or 2)),
                          topology="async", dirichlet=dirichlet or (0.0, 0.0))
    t2 = time.time()
    assert samples_by_game(recs) == oracle_samples_by_game(want)
    print(f"full T3 ok: {blocks}x{channels} bf16 network, {n} games, n_mcts_iterations = {n_iter}, Dirichlet {dirichlet}: ALL {n} games replayed by the oracle from the "
          f"evaluator's own answers, {len(recs)} samples identical; {n_rows} evaluator rows logged ({len(table[0])} distinct positions), {n_dup} repeats of a position, "
          f"every repeat answered with the same bits; device run {t1 - t0:.0f} s, oracle replay {t2 - t1:.0f} s")


if __name__ == "__main__":
    main()
