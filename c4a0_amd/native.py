"""`play_games` through the library's own host loop: `c4_play_games_bf16` (include/c4a0_hip.h, c4a0_amd/csrc/c4_selfplay_host.hip).

The reference's `self_play()` (rust/src/self_play.rs:39-129) is compiled code that owns the whole loop; `c4_play_games_bf16` is
that loop in the library -- sessions, streams, HIP-graph capture and replay, completion polling, tail narrowing, the merged
hand-over of the records -- so that a Rust or C host plays a job with ONE call.  This module is the Python binding of the same
call: it hands the library the device pointers of a `c4a0_amd.nn.InferenceNet` and returns a `PlayGamesResult`.  The Python loop of
`c4a0_amd.api.play_games` / `c4a0_amd.session` implements the same schedule for every other kind of evaluator (numpy callbacks,
arbitrary device callables, tournaments); with an `InferenceNet` the two return the same bytes (tests/test_gpu_native_host.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib
from .results import PlayGamesResult, results_from_records


def network_struct(net) -> _lib.NetworkBf16:
    """c4_network_bf16 over the tensors of a c4a0_amd.nn.InferenceNet (which must stay alive while the struct is used)."""
    import torch

    if not (getattr(net, "path", None) == "hip" and net.dtype == torch.bfloat16 and net.merged_w1 is not None):
        raise TypeError("the native host loop takes a bf16 c4a0_amd.nn.InferenceNet on the hand-written kernels whose heads both have a hidden layer")
    s = _lib.NetworkBf16()
    s.channels, s.n_blocks = net.channels, net.n_blocks
    s.tower_w0, s.tower_w, s.tower_bias = net.tw0.data_ptr(), net.tw.data_ptr(), net.tbias.data_ptr()
    s.w1, s.b1 = net.merged_w1.data_ptr(), net._bias32[net.merged_b1.data_ptr()].data_ptr()
    pol = list(zip(net.pol_w[1:-1], net.pol_b[1:-1]))
    val = list(zip(net.val_w[1:-1], net.val_b[1:-1]))
    if len(pol) > 8 or len(val) > 8:
        raise ValueError("at most 8 further hidden layers per head")
    s.n_policy_hidden, s.n_value_hidden = len(pol), len(val)
    for i, (w, b) in enumerate(pol):
        s.policy_w[i], s.policy_b[i] = w.data_ptr(), net._bias32[b.data_ptr()].data_ptr()
    for i, (w, b) in enumerate(val):
        s.value_w[i], s.value_b[i] = w.data_ptr(), net._bias32[b.data_ptr()].data_ptr()
    s.policy_out_w, s.value_out_w = net.pol_w[-1].data_ptr(), net.val_w[-1].data_ptr()
    s.policy_out_b, s.value_out_b = net.pol_b32.data_ptr(), net.val_b32.data_ptr()
    return s


def _call_interruptibly(call):
    """A job is ONE library call -- seconds to minutes in which the interpreter would not see a signal.  The call runs on a helper
    thread while the caller waits for it (a wait that signals do interrupt): on KeyboardInterrupt (or whatever else arrives) the
    running job is asked to stop (`c4_play_games_cancel`: it returns after the replays in flight, everything given back), and the
    exception goes on to the caller, as it would out of the Python loop.  (The status check runs on the helper thread too: the
    library's error string is per thread.)"""
    import threading

    box = {}

    def work():
        try:
            call()
        except BaseException as e:  # noqa: BLE001  (handed to the caller below)
            box["error"] = e

    t = threading.Thread(target=work, name="c4_play_games_bf16", daemon=True)
    t.start()
    try:
        t.join()
    except BaseException:
        while t.is_alive():
            _lib.lib().c4_play_games_cancel()       # (again and again: a request made before the job's loop started is cleared by it)
            t.join(0.02)
        raise
    if "error" in box:
        raise box["error"]


def run_native(ids: np.ndarray, n_mcts_iterations: int, c_exploration: float, c_ply_penalty: float, net, *, resident_games=None,
               concurrent_sessions=None, steps_per_graph: int = 0, tail_steps_per_graph: int = 0, blocks_per_slot: int = 0, reclaim=None,
               reclaim_period: int = 0, dirichlet=None, eval_cache_entries: int = 0, stats: Optional[dict] = None, on_device: bool = False):
    """One `c4_play_games_bf16` call for the requests `ids` (uint64[n, 3]): (records, counts) in request order; `stats` receives the
    sessions' counters and where the call's wall time went, under the keys the Python loop uses (c4a0_amd/api.py _play).
    on_device=True: the records stay on the GPU, a uint8[n, 64] tensor (what the multi-GPU path all-gathers)."""
    import torch

    from .session import SAMPLE_DTYPE

    n = len(ids)
    ns = network_struct(net)
    opt = _lib.PlayOptions()
    opt.device = net.device.index if net.device.index is not None else torch.cuda.current_device()
    opt.resident_games, opt.concurrent_sessions = int(resident_games or 0), int(concurrent_sessions or 0)
    opt.steps_per_graph, opt.tail_steps_per_graph = int(steps_per_graph), int(tail_steps_per_graph)
    opt.blocks_per_slot, opt.reclaim_period = int(blocks_per_slot), int(reclaim_period)
    opt.flags = _lib.FLAG_RECLAIM if reclaim else (_lib.FLAG_NO_RECLAIM if reclaim is False else 0)
    if dirichlet is not None:
        opt.dirichlet_alpha, opt.dirichlet_epsilon = float(dirichlet[0]), float(dirichlet[1])
    opt.eval_cache_entries = int(eval_cache_entries)
    counts = np.empty(n, dtype=np.uint32)
    cap = n * _lib.MAX_SAMPLES_PER_GAME                                      # 43 per game always suffice
    if on_device:
        recs = torch.empty((cap, 64), dtype=torch.uint8, device=net.device)
        recs_ptr = recs.data_ptr()
    else:
        recs = np.empty(cap, dtype=SAMPLE_DTYPE)                             # (untouched pages are never made resident)
        recs_ptr = recs.ctypes.data
    n_recs, totals, phases = C.c_uint64(), _lib.Counters(), _lib.PlayPhases()
    tab = np.ascontiguousarray(ids, dtype=np.uint64)
    _call_interruptibly(lambda: _lib.check(_lib.lib().c4_play_games_bf16(
        tab.ctypes.data, n, int(n_mcts_iterations), float(c_exploration), float(c_ply_penalty), C.byref(ns), C.byref(opt), counts.ctypes.data, recs_ptr, cap,
        C.byref(n_recs), C.byref(totals), C.byref(phases))))
    if stats is not None:
        ph = phases.as_dict()
        stats.update(totals.as_dict())
        stats.update(steps=ph["rounds"], n_slots=ph["resident_games"], rows_at_end=ph["rows_at_end"], concurrent_sessions=ph["sessions"], host_loop="native")
        stats["phases"] = {"setup_s": ph["setup_s"], "start_and_capture_s": 0.0, "first_capture_inside": "steady_s", "steady_s": ph["steady_s"], "tail_s": ph["tail_s"], "drain_s": ph["drain_s"],
                           "graph_captures": ph["graph_captures"], "recapture_s_inside_steady_and_tail": ph["capture_s"],
                           "rounds_until_all_started": ph["rounds_until_all_started"], "narrowings": []}
    return recs[: n_recs.value], counts


def play_games_native(reqs: Sequence, max_nn_batch_size: int, n_mcts_iterations: int, c_exploration: float, c_ply_penalty: float, net, *,
                      stats: Optional[dict] = None, **options) -> PlayGamesResult:
    """The reference's six arguments (max_nn_batch_size has no meaning in device mode: every resident game's leaf is a row) with
    `net` in place of the callback, and `run_native`'s keywords (`play_games`' plus the graph lengths); the whole job runs inside ONE
    library call.  (`play_games(evaluator=net)` takes this path by itself for an unmodified InferenceNet; this entry point insists.)"""
    from .api import _ids_of

    ids = _ids_of(reqs)
    if bool((ids[:, 1] != ids[:, 2]).any()):
        raise TypeError("games between different models need play_games(evaluator={model_id: evaluator, ...})")
    if int(max_nn_batch_size) < 1 or int(n_mcts_iterations) < 0:
        raise ValueError("max_nn_batch_size must be >= 1 and n_mcts_iterations >= 0")
    if len(ids) == 0:
        return PlayGamesResult([])
    recs, counts = run_native(ids, n_mcts_iterations, c_exploration, c_ply_penalty, net, stats=stats, **options)
    return results_from_records(ids, recs, counts)
