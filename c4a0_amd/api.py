"""`play_games` -- the drop-in entry point (reference rust/src/pybridge.rs:20-53,
src/c4a0_rust/__init__.pyi:49-56).

Same six positional arguments as the reference.  Two evaluator modes:

* callback mode (reference-compatible): `py_eval_pos_cb(model_id, float32[B,2,6,7]) ->
  (float32[B,7], float32[B], float32[B])`, B <= max_nn_batch_size, unique positions per call,
  never called concurrently (pybridge.rs:161-199, self_play.rs:196-237).  Costs a host round
  trip per step -- exactly what the reference does -- while the tree work stays on the GPU.
* device mode (keyword `evaluator=`): a callable on device tensors, `evaluator(planes[G,2,6,7])
  -> (logprobs[G,7], q[G,2])`, e.g. `c4a0_amd.nn.InferenceNet`; leaf batches never leave HBM.

The tree path always runs in the HIP kernels; there is no CPU implementation to fall back to.
"""
from __future__ import annotations

import threading
import time
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from .results import GameMetadata, PlayGamesResult, results_from_records
from .session import DeviceEvaluator, DeviceSession

# Games advanced in lock-step per round when the caller does not say (`resident_games=`).  Two forces (MI355X, BASELINE config 2's
# network, profiles/r06_whole_call.txt): more resident games make a round more efficient (fuller GEMM grids, a step kernel further
# from its latency floor) but the job's TAIL -- the rounds after the last request has been started, when finished slots stay empty
# and a round cannot get shorter than its five launches' latency, ~60 us -- grows with them.  Whole calls, seconds:
#     games    4 096 slots   8 192    16 384    (10 240: 1.633 -- 5 120 rows per session are 1.25 waves of tower workgroups)
#    16 384      0.770       0.770      -
#    40 960      1.603       1.519     1.521
#    81 920      3.048       2.836     2.801
# So: the largest of 4 096 / 8 192 / 16 384 (whole waves of workgroups for every kernel of the evaluator) that the job fills at least
# four times over, and never more tree arena than a quarter of the device's free memory.
DEFAULT_RESIDENT_GAMES = 4096
MAX_DEFAULT_RESIDENT_GAMES = 16384


def default_resident_games(n_games: int, n_mcts_iterations: int, graph_safe: bool, device=None) -> int:
    if n_games <= DEFAULT_RESIDENT_GAMES or not graph_safe:   # callback / multi-model modes: a host round trip per round, rows are not the limit
        return min(n_games, DEFAULT_RESIDENT_GAMES)
    want = DEFAULT_RESIDENT_GAMES
    while want < MAX_DEFAULT_RESIDENT_GAMES and n_games >= 8 * want:
        want *= 2
    # arena bytes per slot (include/c4a0_hip.h c4_config.blocks_per_slot): 43 n + 8 blocks of 128 bytes, or two reclaimed halves above n = 1 000
    n = max(1, n_mcts_iterations)
    per_slot = 128 * ((43 * n + 8) if n <= 1000 else 2 * (5 * n // 2 + 554))
    try:
        free, _total = torch.cuda.mem_get_info(device)
        while want > DEFAULT_RESIDENT_GAMES and want * per_slot > free // 4:
            want //= 2
    except Exception:
        pass
    return want


def _planes_from_bits(mask: np.ndarray, value: np.ndarray) -> np.ndarray:
    """create_pos_batch (pybridge.rs:202-221) for arrays of positions -> float32[B,2,6,7]."""
    bits = np.arange(42, dtype=np.uint64)[None, :]
    p0 = ((value[:, None] >> bits) & np.uint64(1)).astype(np.float32)
    p1 = (((mask & ~value)[:, None] >> bits) & np.uint64(1)).astype(np.float32)
    return np.concatenate([p0, p1], axis=1).reshape(-1, 2, 6, 7)


def _popcount64(x: np.ndarray) -> np.ndarray:
    x = x - ((x >> np.uint64(1)) & np.uint64(0x5555555555555555))
    x = (x & np.uint64(0x3333333333333333)) + ((x >> np.uint64(2)) & np.uint64(0x3333333333333333))
    x = (x + (x >> np.uint64(4))) & np.uint64(0x0F0F0F0F0F0F0F0F)
    return (x * np.uint64(0x0101010101010101)) >> np.uint64(56)


class _CallbackEvaluator:
    """PyEvalPos (pybridge.rs:161-199) + the batching semantics of NNThread::loop_once
    (self_play.rs:196-237): per step, the unique (model, leaf position) pairs of all resident
    games are evaluated in chunks of at most max_nn_batch_size, one model per call.

    The bookkeeping runs on the device (c4_session_unique_leaves: three small launches): the slots
    meet in a hash table, the representatives are ranked in slot order and write their rows of the
    evaluator input straight into a pinned host array -- one PCIe crossing, no staging copy, and
    the host learns the batch size from one pinned word after one stream synchronisation.  The
    answers go back the same way (c4_session_scatter_outputs reads the pinned answer rows and fans
    them out to every slot that asked).  The callback sees exactly what the reference's sees:
    float32 [B, 2, 6, 7], unique positions, B <= max_nn_batch_size, one model per call."""

    def __init__(self, session: DeviceSession, cb: Callable, max_nn_batch_size: int, p0: np.ndarray, p1: np.ndarray):
        import ctypes as C

        from ._lib import check
        self._C, self._check = C, check
        self.s, self.cb, self.cap = session, cb, max(1, int(max_nn_batch_size))
        dev, g = session.device, session.n_slots
        self.models_used = np.unique(np.concatenate([p0, p1]))
        self.multi = self.models_used.size > 1
        self.slot_models = session.bind_leaf_models() if self.multi else None   # the step kernel publishes mcts.rs:70-76 per slot
        self.inverse = torch.zeros(g, dtype=torch.int32, device=dev)            # slot -> row of the batch
        self.h_count = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.h_models = torch.zeros(g, dtype=torch.int64).pin_memory() if self.multi else None
        # pinned staging of the answers on their way back (the input batches are fresh pinned arrays, see __call__)
        self.h_out = torch.empty((g, 9), dtype=torch.float32).pin_memory()
        self.nn_positions = 0

    def __call__(self, _planes: torch.Tensor, fused_step: bool = False):
        s, C = self.s, self._C
        g = s.n_slots
        # A FRESH pinned array per step (PyTorch's caching host allocator: no system call after warm-up), as the
        # reference hands its callback a new array it owns (pybridge.rs:220 into_pyarray): a callback that keeps x
        # keeps this block alive, nothing it was given is ever overwritten.
        h_planes = torch.empty((g, 2, 6, 7), dtype=torch.float32, pin_memory=True)
        self._check(s.L.c4_session_unique_leaves(s._h, C.c_void_p(self.inverse.data_ptr()), C.c_void_p(h_planes.data_ptr()),
                                                 C.c_void_p(self.h_models.data_ptr()) if self.multi else None,
                                                 C.c_void_p(self.h_count.data_ptr())))
        s._bound_stream.synchronize()              # the batch and its size have landed in host memory
        n_u = int(self.h_count[0])
        if n_u == 0:                               # every resident game has finished
            return s.logprobs, s.q
        planes_np, out_np = h_planes.numpy()[:n_u], self.h_out.numpy()
        if self.multi:
            # one model per call (self_play.rs:203-215), ascending ids; ids are 64-bit patterns (>= 2^63 negative here)
            models_h = self.h_models.numpy()[:n_u]
            groups = [(int(m) & ((1 << 64) - 1), np.flatnonzero(models_h == m)) for m in np.unique(models_h.view(np.uint64)).view(np.int64)]
        else:
            groups = [(int(self.models_used[0]), None)]
        for mid, rows in groups:
            n_rows = n_u if rows is None else rows.size
            for i in range(0, n_rows, self.cap):
                j = min(i + self.cap, n_rows)
                sel = slice(i, j) if rows is None else rows[i:j]
                out = self.cb(mid, planes_np[sel])   # a view of the fresh array, or (several models) a gathered copy
                if not (isinstance(out, (tuple, list)) and len(out) == 3):
                    raise TypeError("py_eval_pos_cb must return (policy_logprobs, q_penalty, q_no_penalty)")
                lp, qp, qn = (np.asarray(a) for a in out)
                for name, a, shape in (("policy", lp, (j - i, 7)), ("q_penalty", qp, (j - i,)), ("q_no_penalty", qn, (j - i,))):
                    # pybridge.rs:175-188: contiguous float32 arrays of the batch's shape
                    if a.dtype != np.float32 or a.shape != shape or not a.flags["C_CONTIGUOUS"]:
                        raise TypeError(f"py_eval_pos_cb: {name} must be C-contiguous float32 of shape {shape}, got {a.dtype} {a.shape}")
                out_np[sel, :7], out_np[sel, 7], out_np[sel, 8] = lp, qp, qn
                self.nn_positions += j - i
        # h_out is read by the kernel on the session's stream; the next write to it happens after the next step's
        # synchronisation above, which that kernel precedes
        if fused_step:   # the step kernel itself takes every game's answers from its row (c4_session_step_gather): one launch less
            self._check(s.L.c4_session_step_gather(s._h, C.c_void_p(self.inverse.data_ptr()), C.c_void_p(self.h_out.data_ptr()), n_u))
            return None
        self._check(s.L.c4_session_scatter_outputs(s._h, C.c_void_p(self.inverse.data_ptr()), C.c_void_p(self.h_out.data_ptr()), n_u))
        return s.logprobs, s.q

    gather_step = True   # round(): scatter + step as one launch (False: A/B)

    def round(self):
        """evaluate + step for DeviceSession.round: the answers are handed over inside the step's launch."""
        if not self.gather_step:
            self(None)
            self.s.step()
            return
        if self(None, fused_step=True) is not None:      # every resident game had finished: nothing was evaluated, a plain (idle) step
            self.s.step()


class _MultiModelEvaluator:
    """Device-mode tournaments (reference tournament.py:112-142): the step kernel publishes, per
    slot, the model that must evaluate its leaf (mcts.rs:70-76); the rows are grouped by model on
    the device and every model present evaluates ITS rows only (a gathered batch), the answers are
    scattered back to the slots.  Where the reference serves one model per NN tick
    (self_play.rs:203-215), all leaves are answered every step; a game's trajectory only depends on
    the answers to its own leaves.  Model ids are 64-bit patterns (ids >= 2^63 are negative in the
    int64 tensor; they are matched to the evaluators' keys as unsigned)."""

    def __init__(self, session: DeviceSession, evaluators: dict):
        self.s, self.evs = session, {int(k) & ((1 << 64) - 1): v for k, v in evaluators.items()}
        self.models = session.bind_leaf_models()

    def __call__(self, planes: torch.Tensor):
        lp_out, q_out = self.s.logprobs, self.s.q
        order = torch.argsort(self.models, stable=True)                # slots grouped by model, slot order within a model
        ids, counts = torch.unique_consecutive(self.models[order], return_counts=True)
        lo = 0
        for mid, n in zip(ids.tolist(), counts.tolist()):             # one synchronisation per step
            rows, lo = order[lo:lo + n], lo + n
            ev = self.evs.get(mid & ((1 << 64) - 1))
            if ev is None:
                continue  # idle slots keep id 0 of an absent model
            lp, q = ev(planes.index_select(0, rows))
            lp_out.index_copy_(0, rows, lp.reshape(-1, 7).float())
            q_out.index_copy_(0, rows, q.reshape(-1, 2).float())
        return lp_out, q_out


def _ids_of(reqs) -> np.ndarray:
    """uint64[n, 3] (game_id, player0_id, player1_id) of the requests: the one pass over the caller's objects."""
    try:
        return np.array([(r.game_id, r.player0_id, r.player1_id) for r in reqs], dtype=np.uint64).reshape(-1, 3)
    except AttributeError:
        raise TypeError("reqs must be a sequence of GameMetadata") from None   # reference: extract() fails, pybridge.rs:30


def _validate(reqs, max_nn_batch_size, n_mcts_iterations, py_eval_pos_cb, evaluator, ids=None):
    ids = _ids_of(reqs) if ids is None else ids
    if (py_eval_pos_cb is None) == (evaluator is None):
        raise TypeError("pass exactly one of py_eval_pos_cb (numpy callback) or evaluator= (device callable)")
    if int(max_nn_batch_size) < 1 or int(n_mcts_iterations) < 0:
        raise ValueError("max_nn_batch_size must be >= 1 and n_mcts_iterations >= 0")
    multi = evaluator is not None and isinstance(evaluator, dict)
    if evaluator is not None and not multi and bool((ids[:, 1] != ids[:, 2]).any()):
        raise TypeError("games between different models need evaluator={model_id: evaluator, ...}")
    if multi:
        missing = set(np.unique(ids[:, 1:]).tolist()) - set(evaluator)
        if missing:
            raise KeyError(f"no evaluator for model ids {sorted(missing)}")
    return multi


def play_games(reqs: Sequence[GameMetadata], max_nn_batch_size: int, n_mcts_iterations: int,
               c_exploration: float, c_ply_penalty: float, py_eval_pos_cb: Optional[Callable] = None, *,
               evaluator: Optional[DeviceEvaluator] = None, device=None, resident_games: Optional[int] = None,
               planes_dtype: Optional[torch.dtype] = None, blocks_per_slot: int = 0,
               stats: Optional[dict] = None, dirichlet: Optional[tuple] = None,
               concurrent_sessions: Optional[int] = None, eval_cache_entries: int = 0,
               reclaim: Optional[bool] = None, reclaim_period: int = 0, host_loop: Optional[str] = None) -> PlayGamesResult:
    """Play every game of `reqs` to the end with MCTS self-play on the GPU and return the
    training samples (reference pybridge.rs:20-53).  Results are in `reqs` order (the
    reference's order is thread-finishing order, self_play.rs:116).

    A callback object with a `.device_evaluator` attribute (e.g. `DeviceCallback`) is played in
    device mode with that evaluator: an unmodified caller (training.py:179-189 passes a callable)
    gets the fast path by wrapping its network once, without touching the call.

    Any `n_mcts_iterations` up to 32 200 is accepted, like the reference's heap-allocated tree (mcts.rs:187-206, 332-355): above
    1 000 iterations per move the tree arenas are reclaimed while the games are played (`reclaim=` True / False forces it on /
    off, `DeviceSession`; the samples are the same either way).

    host_loop: who drives the rounds.  With an unmodified bf16 `c4a0_amd.nn.InferenceNet` the whole job runs inside ONE call of the
    library (`c4_play_games_bf16`, csrc/c4_selfplay_host.hip: sessions, the paired HIP graph, polling, narrowing, the merged
    hand-over in C++ -- as the reference's `self_play()` is compiled code, self_play.rs:39-129); every other evaluator (numpy
    callbacks, arbitrary device callables, tournaments, subclasses that override `forward`) is driven by the Python loop of
    c4a0_amd/session.py, which implements the same schedule.  None = that choice; "python" forces the Python loop, "native"
    insists on the library's (TypeError if the evaluator is not one it can run).  Same records either way."""
    reqs = list(reqs)
    if py_eval_pos_cb is not None and evaluator is None and getattr(py_eval_pos_cb, "device_evaluator", None) is not None:
        evaluator, py_eval_pos_cb = py_eval_pos_cb.device_evaluator, None
    metas = _ids_of(reqs)               # u64 ids, as extract() checks (pybridge.rs:30)
    _validate(reqs, max_nn_batch_size, n_mcts_iterations, py_eval_pos_cb, evaluator, metas)
    if not reqs:
        return PlayGamesResult([])
    recs, counts = _play(metas, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty, py_eval_pos_cb, evaluator,
                         device=device, resident_games=resident_games, planes_dtype=planes_dtype, blocks_per_slot=blocks_per_slot, stats=stats,
                         dirichlet=dirichlet, concurrent_sessions=concurrent_sessions, eval_cache_entries=eval_cache_entries, on_device=False,
                         reclaim=reclaim, reclaim_period=reclaim_period, host_loop=host_loop)
    return results_from_records(metas, recs, counts)


def merge_parts(n_games: int, parts):
    """Interleave packed sample records back into request order.

    parts = [(positions int64[k], counts uint32[k], records)] where `records` holds the records of
    requests positions[0], positions[1], ... back to back.  Works on numpy structured arrays
    (SAMPLE_DTYPE) and on torch uint8[n, 64] tensors (any device); vectorised, no per-game loop.
    Returns (records in request order, counts[n_games])."""
    from .session import SAMPLE_DTYPE

    if parts and isinstance(parts[0][2], torch.Tensor):
        dev = parts[0][2].device
        counts = torch.zeros(n_games, dtype=torch.int64, device=dev)
        for pos, c, _ in parts:
            counts[torch.as_tensor(pos, device=dev)] = torch.as_tensor(np.asarray(c).astype(np.int64), device=dev)
        starts = torch.cumsum(counts, 0) - counts
        out = torch.zeros((int(counts.sum().item()), 64), dtype=torch.uint8, device=dev)
        for pos, c, recs in parts:
            c64 = torch.as_tensor(np.asarray(c).astype(np.int64), device=dev)
            src0 = torch.cumsum(c64, 0) - c64
            dst = torch.repeat_interleave(starts[torch.as_tensor(pos, device=dev)] - src0, c64) + torch.arange(recs.shape[0], device=dev)
            out[dst] = recs
        return out, counts.to(torch.int32).cpu().numpy().astype(np.uint32)
    counts = np.zeros(n_games, dtype=np.uint32)
    for pos, c, _ in parts:
        counts[pos] = c
    starts = np.concatenate([[0], np.cumsum(counts, dtype=np.int64)])      # first record of request i in the merged array
    out = np.zeros(int(starts[-1]), dtype=SAMPLE_DTYPE)
    for pos, c, recs in parts:
        c64 = np.asarray(c).astype(np.int64)
        src0 = np.cumsum(c64) - c64
        dst = np.repeat(starts[:-1][pos] - src0, c64) + np.arange(len(recs), dtype=np.int64)
        out[dst] = recs
    return out, counts


def _native_loop_refusal(evaluator, device, planes_dtype, concurrent_sessions) -> Optional[str]:
    """None if `c4_play_games_bf16` can play this job exactly as the Python loop would, else the reason it cannot."""
    from .nn import InferenceNet

    if not isinstance(evaluator, InferenceNet):
        return "the evaluator is not a c4a0_amd.nn.InferenceNet"
    if not evaluator.fused_step_ok:           # (a subclass that overrides forward / __call__ must see every evaluation)
        return "the evaluator overrides InferenceNet.forward, or does not run on the hand-written kernels"
    if evaluator.path != "hip" or evaluator.dtype != torch.bfloat16 or evaluator.merged_w1 is None:
        return "the library's loop takes a bf16 network on the hand-written kernels whose heads both have a hidden layer"
    if len(evaluator.pol_w) - 2 > 8 or len(evaluator.val_w) - 2 > 8:
        return "more than 8 further hidden layers in a head"
    if evaluator.gemm_config != (0, 0) or evaluator.tower_config or not evaluator.use_loader_waves or not evaluator.wide_tiles_r5 or evaluator.stage_hook is not None:
        return "the evaluator carries a measurement switch (tile configuration, stage hook)"
    if concurrent_sessions not in (None, 0, 1, 2):
        return "more than two concurrent sessions"
    if planes_dtype not in (None, torch.bfloat16):
        return "planes_dtype must be bfloat16"
    if device is not None and torch.device(device) != evaluator.device and not (torch.device(device).index is None and torch.device(device).type == "cuda"):
        return "device= names another device than the evaluator's"
    return None


# One job at a time per process: a job captures HIP graphs, and a capture does not tolerate what another job's set-up does meanwhile
# (allocations, transfers on the legacy stream) -- two threads calling play_games at once used to fail with "operation would make the
# legacy stream depend on a capturing stream".  A job fills the device anyway; the second caller waits.  (Re-entrant: a numpy callback
# may itself call play_games.)  The library's own loop holds the same kind of lock for hosts that are not Python (c4_selfplay_host.hip).
_JOB_LOCK = threading.RLock()


def _play(reqs, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty, py_eval_pos_cb, evaluator, device=None,
          resident_games=None, planes_dtype=None, blocks_per_slot=0, stats=None, dirichlet=None, concurrent_sessions=None,
          eval_cache_entries=0, on_device=False, reclaim=None, reclaim_period=0, host_loop=None):
    """Play `reqs` (a uint64[n, 3] table of ids, or GameMetadata-like objects) on ONE device.  Returns (records, counts) in request order; `records` is a numpy
    SAMPLE_DTYPE array, or with on_device=True a uint8[n, 64] tensor that never left the GPU (packed
    by k_pack_samples: what the sample all-gather of the multi-GPU path sends).  Jobs of concurrent threads run one after another."""
    with _JOB_LOCK:
        return _play_locked(reqs, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty, py_eval_pos_cb, evaluator, device=device,
                            resident_games=resident_games, planes_dtype=planes_dtype, blocks_per_slot=blocks_per_slot, stats=stats, dirichlet=dirichlet,
                            concurrent_sessions=concurrent_sessions, eval_cache_entries=eval_cache_entries, on_device=on_device, reclaim=reclaim,
                            reclaim_period=reclaim_period, host_loop=host_loop)


def _play_locked(reqs, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty, py_eval_pos_cb, evaluator, device=None,
                 resident_games=None, planes_dtype=None, blocks_per_slot=0, stats=None, dirichlet=None, concurrent_sessions=None,
                 eval_cache_entries=0, on_device=False, reclaim=None, reclaim_period=0, host_loop=None):
    from .session import run_sessions

    if not isinstance(reqs, np.ndarray):
        reqs = _ids_of(reqs)
    if host_loop not in (None, "native", "python"):
        raise ValueError('host_loop must be None, "native" or "python"')
    if host_loop != "python":
        why = _native_loop_refusal(evaluator, device, planes_dtype, concurrent_sessions)
        if why is None:
            from .native import run_native
            return run_native(reqs, n_mcts_iterations, c_exploration, c_ply_penalty, evaluator, resident_games=resident_games, concurrent_sessions=concurrent_sessions,
                              blocks_per_slot=blocks_per_slot, reclaim=reclaim, reclaim_period=reclaim_period, dirichlet=dirichlet,
                              eval_cache_entries=eval_cache_entries, stats=stats, on_device=on_device)
        if host_loop == "native":
            raise TypeError(f"host_loop='native': {why}")
    multi = evaluator is not None and isinstance(evaluator, dict)
    if planes_dtype is None:   # hand a bf16 network bf16 planes (0/1 are exact): no conversion kernel per step
        planes_dtype = torch.bfloat16 if getattr(evaluator, "dtype", None) == torch.bfloat16 else torch.float32
    graph_safe = evaluator is not None and not multi and getattr(evaluator, "graph_safe", False)
    n_slots = min(len(reqs), int(resident_games) if resident_games else default_resident_games(len(reqs), int(n_mcts_iterations), graph_safe, device))
    # Device evaluators that are pure device code: the resident games are split over sessions that run
    # concurrently on their own streams (session.run_sessions), two by default when each half still
    # fills the GEMMs.  Which session plays a game does not change its samples.
    # ... unless the job is ONE generation (no more requests than slots: nothing is ever refilled, the whole job is its tail): below
    # 2 048 rows a single chain's round is shorter than two paired chains' (57 against 65 us at 1 024 games, profiles/r06_whole_call.txt
    # (d)), which outweighs the full-width rounds a pair plays faster (104 against 120 us at 4 096): 4 096 games on 4 096 slots take
    # 0.280 s with one session, 0.297 with two (40 960 games on 4 096 slots: 1.77 against 1.62 -- there the pair wins).
    # (That is the 32-channel network; with the 64-channel one, five times the arithmetic per row, the pair still wins or ties for one
    # generation: 8 192 games at n = 200 in 2.79 against 2.89 s, 4 096 at n = 800 in 6.20 against 6.23.)
    one_generation = len(reqs) <= n_slots and getattr(evaluator, "channels", 32) <= 32
    parts = int(concurrent_sessions) if concurrent_sessions else (2 if graph_safe and n_slots >= 2048 and not one_generation else 1)
    if parts > 1 and not graph_safe:
        raise TypeError("concurrent_sessions > 1 needs a graph-safe device evaluator (c4a0_amd.nn.InferenceNet)")
    parts = max(1, min(parts, n_slots))
    if eval_cache_entries and multi:
        raise TypeError("eval_cache_entries needs ONE evaluator (not evaluator={model_id: ...})")
    # Rounds per HIP-graph replay.  Long graphs amortise the replay boundary (BASELINE config 2, two sessions: 0.109 ms per round at
    # 8 rounds per replay, 0.104 at 64) but the job's end is only noticed between replays, and up to three may be in flight.  So: long
    # graphs while the job is long -- judged by the rounds it will take, ~15 moves x n simulations per generation of games -- and,
    # for two paired sessions, short ones from the first narrowing of the tail on, where the graph is captured again anyway
    # (session._run_pair).  The reference's default job (1 700 games, n = 1 400: 37 000 rounds, one session) replays 32.
    est_rounds = -(-len(reqs) // max(1, n_slots)) * 15 * max(1, int(n_mcts_iterations))
    if not graph_safe:
        steps_per_graph = tail_steps_per_graph = 0
    elif parts == 2:
        steps_per_graph = 64 if est_rounds >= 4000 else (32 if est_rounds >= 1500 else 8)
        tail_steps_per_graph = 16 if steps_per_graph >= 32 else 8
    elif est_rounds >= 10000:     # one session, a long job (the reference's default job: 37 000 rounds): the replay boundary matters all the way
        steps_per_graph = tail_steps_per_graph = 32
    else:                         # one session, e.g. one generation of config 2's games (3 100 rounds)
        steps_per_graph = 32 if est_rounds >= 1500 else 8
        tail_steps_per_graph = 16 if steps_per_graph >= 32 else 8
    if hasattr(evaluator, "latency_mode"):   # InferenceNet: tile choice of the narrow layers, alone vs beside another session
        evaluator.latency_mode = parts == 1
    phases = {} if stats is not None else None
    t_play0 = time.perf_counter()
    sessions = []
    try:
        for p in range(parts):   # session p plays requests p, p + parts, ...
            mine = reqs[p::parts]
            slots = min(len(mine), (n_slots + parts - 1 - p) // parts)
            s = DeviceSession(max(1, slots), n_mcts_iterations, c_exploration, c_ply_penalty, device=device,
                              planes_dtype=planes_dtype, blocks_per_slot=blocks_per_slot, reclaim=reclaim, reclaim_period=reclaim_period)
            sessions.append(s)
            s.set_games(mine)
            if dirichlet is not None:   # extension: (alpha, epsilon) root noise; the reference has none
                s.set_dirichlet(*dirichlet)
            if eval_cache_entries:      # extension: evaluation cache, each session keeps its own table
                s.set_eval_cache(max(1024, int(eval_cache_entries) // parts))
        t_run0 = time.perf_counter()
        if parts > 1:
            steps = max(run_sessions(sessions, evaluator, steps_per_graph=steps_per_graph, phases=phases, tail_steps_per_graph=tail_steps_per_graph))
        elif evaluator is None:
            ev = _CallbackEvaluator(sessions[0], py_eval_pos_cb, max_nn_batch_size, reqs[:, 1], reqs[:, 2])
            steps = sessions[0].run(ev, poll_every=4, phases=phases)   # the completion probe every 4th step: at most 3 idle steps at the very end
        elif multi:
            steps = sessions[0].run(_MultiModelEvaluator(sessions[0], evaluator), phases=phases)
        else:
            # a c4a0_amd.nn.InferenceNet is pure device code: replay it and the step kernel from a HIP graph
            steps = sessions[0].run(evaluator, steps_per_graph=steps_per_graph, phases=phases, tail_steps_per_graph=tail_steps_per_graph)
        t_drain0 = time.perf_counter()
        pieces = []
        merge_on_device = on_device or parts > 1   # several sessions: interleave their records on the device, ONE transfer to the host
        for p, s in enumerate(sessions):
            pos = np.arange(p, len(reqs), parts, dtype=np.int64)
            pieces.append((pos, s.sample_counts(), s.pack_samples_device() if merge_on_device else s.drain_samples()))
        if parts == 1:
            recs, counts = pieces[0][2], pieces[0][1]
        else:
            recs, counts = merge_parts(len(reqs), pieces)
        if merge_on_device and not on_device:
            from .session import SAMPLE_DTYPE
            recs = recs.cpu().numpy().reshape(-1).view(SAMPLE_DTYPE)
        if stats is not None:
            t_end = time.perf_counter()
            # where the call's wall time went (seconds): session set-up, graph captures, the rounds until every request had been
            # started ("steady": all slots busy), the rounds after that ("tail": finished slots stay empty), the sample hand-over
            t_loop0 = phases.get("t_loop0", t_run0)
            t_started = phases.get("t_all_started", t_drain0)
            stats["phases"] = {"setup_s": t_run0 - t_play0, "start_and_capture_s": t_loop0 - t_run0, "steady_s": t_started - t_loop0,
                               "tail_s": t_drain0 - t_started, "drain_s": t_end - t_drain0, "graph_captures": phases.get("captures", 0),
                               "recapture_s_inside_steady_and_tail": phases.get("recapture_s", 0.0),
                               "rounds_until_all_started": phases.get("steps_all_started"), "narrowings": phases.get("narrowings", [])}
            tot = {}
            for s in sessions:
                for k, v in s.counters().items():
                    tot[k] = tot.get(k, 0) + v if k not in ("error", "error_slot") else max(tot.get(k, 0), v)
            stats.update(tot)
            stats["steps"] = steps
            stats["n_slots"] = sum(s.n_slots for s in sessions)
            stats["rows_at_end"] = sum(s.rows for s in sessions)
            stats["concurrent_sessions"] = parts
            stats["host_loop"] = "python"
            if evaluator is None:
                stats["nn_positions"] = ev.nn_positions
    finally:
        for s in sessions:
            s.close()
    return recs, counts


def trim_cached_memory() -> None:
    """Give the tree arena that the library keeps for the next session (c4_trim_cached_memory, include/c4a0_hip.h)
    back to the device: for callers that want every byte of HBM between two self-play phases."""
    from ._lib import check, lib

    check(lib().c4_trim_cached_memory())


class DeviceCallback:
    """A `py_eval_pos_cb`-shaped object for unmodified callers (training.py:179-189 builds
    `lambda model_id, x: model.forward_numpy(x)`): calling it answers numpy batches exactly like the
    reference callback, and `play_games` recognises `.device_evaluator` and plays in device mode, so the
    leaf batches never leave HBM.  `evaluator` is a DeviceEvaluator or {model_id: DeviceEvaluator}."""

    def __init__(self, evaluator, device=None):
        self.device_evaluator = evaluator
        self.device = device

    def __call__(self, model_id: int, x: np.ndarray):
        ev = self.device_evaluator[model_id] if isinstance(self.device_evaluator, dict) else self.device_evaluator
        dev = self.device if self.device is not None else getattr(ev, "device", None)
        with torch.no_grad():
            lp, q = ev(torch.from_numpy(np.ascontiguousarray(x)).to(dev))
            lp, q = lp.float().cpu().numpy(), q.float().cpu().numpy()
        return np.ascontiguousarray(lp), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])


def run_tui(*_a, **_k):
    """reference pybridge.rs:231-251: terminal UI -- out of scope."""
    raise NotImplementedError("run_tui (interactive terminal UI) is out of scope of the self-play generator")
