"""Every BASELINE.json configuration (and bench.py's own configuration) through the HIP path against
the oracle, at full size.

The oracle cannot replay 4 096 games x 10 000 simulations inside the suite's budget, so each test
checks EVERY game structurally (size-independent properties of the domain, tests/test_gpu_full_size.py)
and replays a subset of the games in the oracle, bit for bit:
  * hash evaluator (an exact integer function of the position): the oracle runs the same games;
  * bf16 network: T3 replay -- the run logs every (leaf -> evaluator output) pair the
    subset's slots consumed, the oracle replays those games with a lookup evaluator.
Reference: rust/src/self_play.rs:268-323, mcts.rs:83-108."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


# ---------------------------------------------------------------------------------------------
def _net(blocks, channels, seed=1337):
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(seed)
    return InferenceNet(ConnectFourNet(ModelConfig(blocks, channels, 4, 2)), torch.device("cuda:0"), dtype=torch.bfloat16)


def _run_logged(net, ids, n_slots, n_iter, log_slots, dirichlet=None):
    """One eager session; every step, the evaluator rows of `log_slots` (input planes, outputs) are
    appended to a device-side log.  Returns (records, counts, counters, log) with log[slot] = the
    sequence of (leaf position, outputs bytes) that slot's evaluator row went through, one per step."""
    from c4a0_amd.session import DeviceSession
    from tests.helpers import planes_to_pos_np

    dev = torch.device("cuda:0")
    s = DeviceSession(n_slots, n_iter, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games([(g, 0, 0) for g in ids])
    if dirichlet is not None:
        s.set_dirichlet(*dirichlet)
    idx = torch.as_tensor(np.asarray(log_slots, dtype=np.int64), device=dev)
    log_p, log_lp, log_q = [], [], []

    def log(_step):   # planes hold the leaves, logprobs/q the evaluator's answers for them
        log_p.append(s.planes.index_select(0, idx))
        log_lp.append(s.logprobs.index_select(0, idx))
        log_q.append(s.q.index_select(0, idx))

    s.run(net, on_step=log, poll_every=64)
    recs, counts, ctr = s.drain_samples(), s.sample_counts(), s.counters()
    s.close()
    k = len(log_slots)
    planes = torch.stack(log_p).float().cpu().numpy()                     # [steps, k, 2, 6, 7]
    lp, q = torch.stack(log_lp).cpu().numpy(), torch.stack(log_q).cpu().numpy()
    mask, value = planes_to_pos_np(planes.reshape(-1, 2, 6, 7))
    mask, value = mask.reshape(-1, k), value.reshape(-1, k)
    seq = {}
    for j, slot in enumerate(log_slots):
        seq[slot] = [((int(m), int(v)), a.tobytes(), b.tobytes()) for m, v, a, b in zip(mask[:, j], value[:, j], lp[:, j], q[:, j])]
    return recs, counts, ctr, seq


def _oracle_replay(seq_by_slot, slot_ids, n_iter, dirichlet=(0.0, 0.0)):
    """T3: the oracle plays each logged game ALONE and is answered, leaf by leaf, with
    what the device's evaluator said for that game's row at that step.  The k-th non-terminal leaf
    the oracle asks about must BE the k-th leaf the device showed its evaluator (checked), so the
    whole search sequence is compared, not only the samples -- and every logged position must have been
    answered with the same bits wherever and whenever it was shown (the evaluator is a function of the
    position; round 2's library GEMMs were not).
    seq_by_slot[slot] as returned by _run_logged; slot_ids = [(slot, game id)]."""
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game, planes_to_pos_np

    # The evaluator is a FUNCTION OF THE POSITION (round 3: hand-written GEMMs with one fixed summation
    # order per element): whichever slot, row or step showed a position, the answer has the same bits.
    # (Not with Dirichlet noise off/on: noise does not touch the evaluator.)  Checked over every logged row.
    table = {}
    for seq in seq_by_slot.values():
        for pos, a, b in seq:
            if table.setdefault(pos, (a, b)) != (a, b):
                raise AssertionError(f"the evaluator answered position {pos} with different bits in different rows / steps")

    out = {}
    for slot, gid in slot_ids:
        seq, cursor = seq_by_slot[slot], [0]

        def lookup(_model_id, x, seq=seq, cursor=cursor):
            mask, value = planes_to_pos_np(x)
            assert len(mask) == 1
            pos = (int(mask[0]), int(value[0]))
            if cursor[0] < len(seq) and seq[cursor[0]][0] == pos:
                _p, a, b = seq[cursor[0]]
                cursor[0] += 1
                lp, q = np.frombuffer(a, dtype=np.float32), np.frombuffer(b, dtype=np.float32)
                return lp.reshape(1, 7).copy(), q[:1].copy(), q[1:2].copy()
            # the device never shows a terminal leaf to the evaluator when it can run that simulation in
            # the launch that selected it; the reference asks and ignores the answer (mcts.rs:92-98)
            assert O.terminal_state(O.Pos(*pos)) != 0, f"game {gid}: leaf {pos} is not the device's next leaf {seq[cursor[0]][0] if cursor[0] < len(seq) else None}"
            return np.zeros((1, 7), np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32)

        want, _ = O.self_play([(gid, 0, 0)], 64, n_iter, 6.6, 0.01, lookup, dirichlet=dirichlet)
        out.update(oracle_samples_by_game(want))
    return out


_COL0 = 0x810204081
_COL_UP_TO = np.array([_COL0 & ((1 << (7 * h)) - 1) for h in range(8)], dtype=np.uint64)   # column 0's cells below height h


def _keys_to_positions(keys):
    """c4_session_leaf_keys' int64 (value bits | 7 column heights << 42) -> (mask, value) uint64 arrays."""
    k = keys.astype(np.uint64)
    value = k & np.uint64((1 << 42) - 1)
    mask = np.zeros_like(value)
    for c in range(7):
        h = (k >> np.uint64(42 + 3 * c)) & np.uint64(7)
        mask |= _COL_UP_TO[h.astype(np.int64)] << np.uint64(c)
    return mask, value


def _run_logging_every_row(net, ids, n_slots, n_iter, dirichlet=None):
    """One eager session; at EVERY step the key of every slot's leaf and the evaluator's answer for it are logged on the device.
    Returns (records, counts, counters, table) with table = (mask, value, out[n, 9]) sorted by (mask, value): what the evaluator
    said for every distinct position it was shown during the whole job -- after checking that it said the SAME bits every time it
    was shown a position again (the evaluator is a function of the position), over every row of every step."""
    import ctypes as C
    from c4a0_amd._lib import check
    from c4a0_amd.session import DeviceSession

    dev = torch.device("cuda:0")
    s = DeviceSession(n_slots, n_iter, 6.6, 0.01, device=dev, planes_dtype=torch.bfloat16)
    s.set_games([(g, 0, 0) for g in ids])
    if dirichlet is not None:
        s.set_dirichlet(*dirichlet)
    log_k, log_o = [], []

    def log(_step):
        keys = torch.empty(n_slots, dtype=torch.int64, device=dev)
        check(s.L.c4_session_leaf_keys(s._h, C.c_void_p(keys.data_ptr())))
        log_k.append(keys)
        log_o.append(torch.cat([s.logprobs, s.q], dim=1))      # [n_slots, 9], a copy

    s.run(net, on_step=log, poll_every=64)
    recs, counts, ctr = s.drain_samples(), s.sample_counts(), s.counters()
    s.close()
    keys = torch.stack(log_k).reshape(-1)
    out = torch.stack(log_o).reshape(-1, 9)
    live = keys >= 0
    keys, out = keys[live], out[live]
    order = torch.argsort(keys, stable=True)
    keys, out = keys[order], out[order]
    dup = keys[1:] == keys[:-1]
    same_bits = (out.view(torch.int32)[1:][dup] == out.view(torch.int32)[:-1][dup]).all()
    assert bool(same_bits), "the evaluator answered one position with different bits in different rows / steps"
    first = torch.ones_like(keys, dtype=torch.bool)
    first[1:] = ~dup
    n_rows, n_dup = int(keys.numel()), int(dup.sum())
    keys_u, out_u = keys[first].cpu().numpy(), out[first].cpu().numpy()
    mask, value = _keys_to_positions(keys_u)
    o = np.lexsort((value, mask))
    return recs, counts, ctr, (mask[o], value[o], np.ascontiguousarray(out_u[o])), (n_rows, n_dup)


def test_config2_every_game_of_a_generation_replayed_by_the_oracle_with_the_networks_own_answers():
    """T3 at FULL size (round 6; subsets of 24-96 games before): BASELINE config 2's 4 096 games with the real bf16
    4 x 32 network, one generation.  The run logs what the evaluator answered for every row of every step (6 M rows); those
    answers, keyed by position, become the oracle's evaluator (c4o_eval_table: a position it was never shown and that is not
    terminal fails the replay), and EVERY game's samples must equal the oracle's bit for bit.  On the way: every position the
    evaluator saw more than once got the same bits every time."""
    from oracle import c4oracle as O
    from tests.helpers import evidence, oracle_samples_by_game, samples_by_game
    from tests.test_gpu_full_size import _check_structure

    n, n_iter = 4096, 100
    ids = list(range(n))
    recs, counts, ctr, table, (n_rows, n_dup) = _run_logging_every_row(_net(4, 32), ids, n, n_iter)
    assert ctr["games_done"] == n and ctr["error"] == 0
    _check_structure(recs, counts, ids)
    want, ost = O.self_play([(g, 0, 0) for g in ids], 4096, n_iter, 6.6, 0.01, ("table",) + table,
                            n_threads=max(2, min(16, os.cpu_count() or 2)), topology="async")
    assert samples_by_game(recs) == oracle_samples_by_game(want)
    evidence(f"config 2 with the bf16 4x32 network (4 096 games, n = 100): ALL {n} games replayed by the oracle from the evaluator's own answers (T3): "
             f"{len(recs)} samples identical; {n_rows} evaluator rows logged, {n_dup} of them repeats of a position, every repeat answered with the same bits")


def _subset(recs, sub):
    from tests.helpers import samples_by_game

    return samples_by_game(recs[np.isin(recs["game_id"], np.array(sub, dtype=np.uint64))])


def _hash_run(ids, n_slots, n_iter):
    from c4a0_amd.session import DeviceSession
    from tests.helpers import GraphSafeHashEval

    s = DeviceSession(n_slots, n_iter, 6.6, 0.01, planes_dtype=torch.bfloat16)
    s.set_games([(g, 0, 0) for g in ids])
    s.run(GraphSafeHashEval(), steps_per_graph=16)
    recs, counts, ctr = s.drain_samples(), s.sample_counts(), s.counters()
    s.close()
    return recs, counts, ctr


# ---------------------------------------------------------------------------------------------
def test_config4_4096_games_n800_hash_evaluator_structure_and_oracle_subset():
    """BASELINE config 4's tree shape: 4 096 concurrent games, n_mcts_iterations = 800 (deep trees,
    the largest arenas, paths beyond 16 levels)."""
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game
    from tests.test_gpu_full_size import _check_structure

    n, n_iter = 4096, 800
    ids = list(range(70000, 70000 + n))
    recs, counts, ctr = _hash_run(ids, n, n_iter)
    assert ctr["games_done"] == n and ctr["error"] == 0 and ctr["samples"] == len(recs) == counts.sum()
    _check_structure(recs, counts, ids)
    assert ctr["sims"] / n > 7 * 300
    sub = sorted(np.random.default_rng(4).choice(ids, 1024, replace=False).tolist())   # round 6: 1 024 games through the oracle (128 in round 5, 16 before): 10 M simulations
    want, _ = O.self_play([(g, 0, 0) for g in sub], 4096, n_iter, 6.6, 0.01, "hash", n_threads=max(2, min(16, os.cpu_count() or 2)), topology="async")
    assert _subset(recs, sub) == oracle_samples_by_game(want)
    from tests.helpers import evidence
    evidence(f"config 4 tree shape (4 096 games, n = 800, hash evaluator): {len(recs)} samples structurally checked, {len(sub)} games == oracle bit for bit")


def test_config4_4096_games_n800_8x64_network_t3_replay():
    """BASELINE config 4 itself: 4 096 games, n = 800, 8-block / 64-channel bf16 ResNet (4 policy and
    2 value layers), T3 replay of a subset by the oracle."""
    from tests.test_gpu_full_size import _check_structure

    n, n_iter = 4096, 800
    ids = list(range(n))
    log_slots = sorted(np.random.default_rng(44).choice(n, 32, replace=False).tolist())   # no refill: slot g plays game g (round 5: 32 games replayed, 12 before)
    recs, counts, ctr, seq = _run_logged(_net(8, 64), ids, n, n_iter, log_slots)
    assert ctr["games_done"] == n and ctr["error"] == 0
    _check_structure(recs, counts, ids)
    assert _subset(recs, log_slots) == _oracle_replay(seq, [(g, ids[g]) for g in log_slots], n_iter)
    from tests.helpers import evidence
    evidence(f"config 4 (4 096 games, n = 800, 8x64 bf16 network): {len(recs)} samples structurally checked, {len(log_slots)} games replayed leaf by leaf by the oracle (T3)")


@pytest.mark.parametrize("dirichlet", [None, (0.3, 0.25)])
def test_config5_per_rank_8192_games_n200_8x64_network_plain_and_dirichlet(dirichlet):
    """BASELINE config 5, one rank's share: 8 192 games, n = 200, 8 x 64 network, temperature schedule
    (always on: self_play.rs:294-299), without and with Dirichlet root noise (extension; the oracle's
    c4o_dirichlet is its specification).  Game ids follow the rank-3-of-8 shard pattern."""
    from tests.test_gpu_full_size import _check_structure

    n, n_iter = 8192, 200
    ids = [3 + 8 * i for i in range(n)]
    log_slots = sorted(np.random.default_rng(5).choice(n, 48, replace=False).tolist())   # round 5: 48 games replayed (16 before)
    recs, counts, ctr, seq = _run_logged(_net(8, 64), ids, n, n_iter, log_slots, dirichlet=dirichlet)
    assert ctr["games_done"] == n and ctr["error"] == 0
    _check_structure(recs, counts, ids)
    sub = [ids[g] for g in log_slots]
    assert _subset(recs, sub) == _oracle_replay(seq, [(g, ids[g]) for g in log_slots], n_iter, dirichlet or (0.0, 0.0))
    from tests.helpers import evidence
    evidence(f"config 5 per-rank share (8 192 games, n = 200, 8x64 bf16 network, Dirichlet {dirichlet}): {len(recs)} samples structurally checked, {len(sub)} games replayed by the oracle (T3)")


def test_config3_shard_pattern_equals_the_single_rank_run():
    """BASELINE config 3: 32 768 games sharded over 8 ranks, id = r + 8 i, 4 096 resident per rank.
    Ranks 0 and 5 of that pattern, each played alone, give exactly the records those games get when
    ONE session plays all 32 768 (which slot, session or rank plays a game changes nothing)."""
    from oracle import c4oracle as O
    from tests.helpers import oracle_samples_by_game

    n_all, world, n_iter = 32768, 8, 100
    all_ids = list(range(n_all))
    full, full_counts, ctr = _hash_run(all_ids, 4096, n_iter)
    assert ctr["games_done"] == n_all and ctr["error"] == 0
    offs = np.concatenate([[0], np.cumsum(full_counts.astype(np.int64))])
    for r in (0, 5):
        ids = all_ids[r::world]
        recs, counts, c = _hash_run(ids, 4096, n_iter)
        assert c["games_done"] == len(ids)
        assert np.array_equal(counts, full_counts[r::world])
        want = np.concatenate([full[offs[g]:offs[g + 1]] for g in ids])
        assert recs.tobytes() == want.tobytes()
    sub = [5 + 8 * i for i in (0, 17, 900, 4095)]
    ora, _ = O.self_play([(g, 0, 0) for g in sub], 64, n_iter, 6.6, 0.01, "hash")
    assert _subset(full, sub) == oracle_samples_by_game(ora)


def test_bench_configuration_graph_two_sessions_equals_eager_and_oracle():
    """bench.py's configuration: BASELINE config 2's network (4 blocks x 32 channels, 4 policy / 2
    value layers, bf16), n = 100, 4 096 resident games as TWO concurrent sessions of 2 048 replaying
    HIP graphs, slots refilled from the queue (8 192 games).  Session p plays requests p, p + 2, ...;
    each session's games are also played by an EAGER session of the same width (same evaluator batch
    shape), one of them logged and replayed by the oracle (T3) on a subset; the graph-replayed
    two-session job must return exactly the two eager runs' records, merged."""
    import c4a0_amd
    from c4a0_amd.api import merge_parts

    n, n_iter = 8192, 100
    net = _net(4, 32)
    ids = list(range(n))
    log_slots = sorted(np.random.default_rng(2).choice(2048, 96, replace=False).tolist())   # first generation: slot g plays the session's game g (round 5: 96 games replayed, 24 before)
    ids0, ids1 = ids[0::2], ids[1::2]
    recs0, counts0, ctr0, seq = _run_logged(net, ids0, 2048, n_iter, log_slots)
    assert ctr0["games_done"] == len(ids0) and ctr0["error"] == 0
    assert _subset(recs0, [ids0[g] for g in log_slots]) == _oracle_replay(seq, [(g, ids0[g]) for g in log_slots], n_iter)
    recs1, counts1, ctr1, _ = _run_logged(net, ids1, 2048, n_iter, [0])
    want, want_counts = merge_parts(n, [(np.arange(0, n, 2), counts0, recs0), (np.arange(1, n, 2), counts1, recs1)])
    st = {}
    res = c4a0_amd.play_games([c4a0_amd.GameMetadata(g, 0, 0) for g in ids], 4096, n_iter, 6.6, 0.01, evaluator=net,
                              resident_games=4096, concurrent_sessions=2, stats=st)
    assert st["concurrent_sessions"] == 2 and st["n_slots"] == 4096 and st["games_done"] == n
    got, got_counts = res.to_records()
    assert np.array_equal(got_counts, want_counts)
    assert got.tobytes() == want.tobytes()
    from tests.helpers import evidence
    evidence(f"bench configuration (config 2 network, 2 x 2 048 slots, 8 192 games): paired-graph play_games == two eager sessions byte for byte ({len(got)} samples), {len(log_slots)} games replayed by the oracle (T3)")
