"""Host driver of one GPU's self-play session.

Mirrors the role of `self_play()` (reference rust/src/self_play.rs:39-129): it owns the loop
"evaluate the leaves of all games -> give every game its MCTS job" until all games are over.
Here the MCTS jobs of all resident games are one HIP kernel (`c4_session_step`) and the
evaluator runs on the same stream on persistent tensors, so leaf batches never leave HBM.
PyTorch supplies device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import C4Error, Config, Counters, GameMetadataC, SampleRec, check

DeviceEvaluator = Callable[[torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]
"""evaluator(planes[G,2,6,7]) -> (policy_logprobs[G,7] float32, q[G,2] float32) on the same device."""

# Stream captures are opened in hipStreamCaptureModeThreadLocal: only THIS thread's calls are checked against the
# open capture.  In the default (global) mode a capture-unsafe call from ANY thread -- RCCL's watchdog thread polls
# its events with hipEventQuery for as long as a process group lives -- invalidates it, and a training loop calls
# play_games_sharded with the group up and collectives just issued (tests/test_gpu_sharded.py).
CAPTURE_ERROR_MODE = "thread_local"


class _capture:
    """`torch.cuda.graph(graph, stream=..., capture_error_mode=...)` without its `torch.cuda.empty_cache()`: a job re-captures its
    graph at every narrowing of its tail (5-7 captures per job), and handing every cached block back to the driver before each one
    -- and allocating the capture's activations afresh -- is most of what a capture costs (profiles/r06_whole_call.txt (e)).
    The callers synchronise the device themselves before they capture."""

    def __init__(self, graph: "torch.cuda.CUDAGraph", stream: Optional[torch.cuda.Stream] = None):
        self.graph = graph
        self.stream = stream if stream is not None else torch.cuda.Stream()   # a capture cannot run on the legacy default stream
        self.ctx = torch.cuda.stream(self.stream)

    def __enter__(self):
        self.ctx.__enter__()
        self.graph.capture_begin(capture_error_mode=CAPTURE_ERROR_MODE)

    def __exit__(self, *args):
        self.graph.capture_end()
        self.ctx.__exit__(*args)


def capture(graph: "torch.cuda.CUDAGraph", stream: Optional[torch.cuda.Stream] = None):
    return _capture(graph, stream) if LEAN_CAPTURE else torch.cuda.graph(graph, stream=stream, capture_error_mode=CAPTURE_ERROR_MODE)


LEAN_CAPTURE = True   # (False: torch.cuda.graph's own entry, the A/B)


class DeviceSession:
    def __init__(self, n_slots: int, n_mcts_iterations: int, c_exploration: float, c_ply_penalty: float,
                 device: Optional[torch.device] = None, planes_dtype: torch.dtype = torch.float32,
                 blocks_per_slot: int = 0, no_moves: bool = False, one_sim_per_step: bool = False,
                 reclaim: Optional[bool] = None, reclaim_period: int = 0):
        """reclaim: True = the tree arena is reclaimed while games are played (C4_FLAG_RECLAIM: the live subtree is copied into the
        arena's other half when one runs short, as the reference frees dead subtrees at every move, mcts.rs:187-206); None = the
        library decides (on above 1 000 iterations per move with the default sizing); False with more than 1 523 iterations needs
        an explicit blocks_per_slot.  reclaim_period: step launches between two looks at the arenas (0 = 64; tests use 1)."""
        if not torch.cuda.is_available():
            raise RuntimeError("c4a0_amd needs a HIP device: the tree kernels have no CPU fallback")
        self.L = _lib.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        if self.device.type != "cuda":
            raise ValueError("device must be a cuda (HIP) device")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        if planes_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("planes_dtype must be float32 or bfloat16")
        self.n_slots = int(n_slots)
        self.n_mcts_iterations = int(n_mcts_iterations)
        cfg = Config(self.n_slots, int(blocks_per_slot), self.n_mcts_iterations, float(c_exploration),
                     float(c_ply_penalty), 0 if planes_dtype == torch.float32 else 1,
                     (_lib.FLAG_NO_MOVES if no_moves else 0) | (_lib.FLAG_ONE_SIM_PER_STEP if one_sim_per_step else 0) |
                     (_lib.FLAG_RECLAIM if reclaim else (_lib.FLAG_NO_RECLAIM if reclaim is False else 0)), dev_index, int(reclaim_period))
        h = C.c_void_p()
        check(self.L.c4_session_create(C.byref(cfg), C.byref(h)))
        self._h = h
        with torch.cuda.device(self.device):
            self.planes = torch.zeros((self.n_slots, 2, 6, 7), dtype=planes_dtype, device=self.device)
            self.logprobs = torch.zeros((self.n_slots, 7), dtype=torch.float32, device=self.device)
            self.q = torch.zeros((self.n_slots, 2), dtype=torch.float32, device=self.device)
        self.n_games = 0
        self.rows = self.n_slots          # slots a step launches / rows the evaluator computes (compact() narrows it)
        self._bound_stream = None
        self._timing, self._extensions = True, False   # what c4_session_step_head_out refuses: per-launch timing, noise / cache

    # ---------------------------------------------------------------- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self.L.c4_session_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---------------------------------------------------------------- setup
    def set_games(self, reqs, start_positions: Optional[Sequence[Tuple[int, int]]] = None):
        """reqs: (game_id, player0_id, player1_id) triples, or a uint64[n, 3] array of them (no per-game Python work)."""
        if isinstance(reqs, np.ndarray):
            tab = np.ascontiguousarray(reqs, dtype=np.uint64).reshape(-1, 3)
        else:
            tab = np.array([(int(g), int(a), int(b)) for g, a, b in reqs], dtype=np.uint64).reshape(-1, 3)
        n = len(tab)
        arr = C.cast(tab.ctypes.data, C.POINTER(GameMetadataC)) if n else (GameMetadataC * 1)()
        sm = sv = None
        if start_positions is not None:
            if len(start_positions) != n:
                raise ValueError("start_positions must match reqs")
            sm = (C.c_uint64 * max(1, n))(*[int(m) for m, _ in start_positions])
            sv = (C.c_uint64 * max(1, n))(*[int(v) for _, v in start_positions])
        check(self.L.c4_session_set_games(self._h, arr, n, sm, sv))   # copies the list (c4a0_hip.h): `tab` may go
        self.n_games = n
        self.rows = self.n_slots

    def bind(self, stream: Optional[torch.cuda.Stream] = None):
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        check(self.L.c4_session_bind_io(self._h, self.planes.data_ptr(), self.logprobs.data_ptr(), self.q.data_ptr(),
                                        C.c_void_p(st.cuda_stream)))
        self._bound_stream = st

    def set_dirichlet(self, alpha: float, epsilon: float):
        """Extension (not in the reference): Dirichlet noise on the priors of every search root."""
        check(self.L.c4_session_set_dirichlet(self._h, float(alpha), float(epsilon)))
        self._extensions = self._extensions or float(epsilon) > 0.0

    def set_eval_cache(self, n_entries: int, max_sims_per_step: int = 0):
        """Extension (not in the reference, off by default): keep the evaluator's outputs by position in a
        direct-mapped table of `n_entries` x 64 bytes in HBM; a game whose new leaf is found there runs
        that simulation in the same launch instead of using an evaluator row (c4_session_set_eval_cache).
        Needs an evaluator that is a deterministic function of the position.  Call before start()."""
        check(self.L.c4_session_set_eval_cache(self._h, int(n_entries), int(max_sims_per_step)))
        self._extensions = self._extensions or int(n_entries) > 0

    def bind_leaf_models(self) -> torch.Tensor:
        """int64[n_slots] tensor that start()/step() fill with the model id to evaluate each leaf with."""
        self.leaf_models = torch.zeros(self.n_slots, dtype=torch.int64, device=self.device)
        check(self.L.c4_session_bind_leaf_models(self._h, C.c_void_p(self.leaf_models.data_ptr())))
        return self.leaf_models

    def arena(self) -> dict:
        """How the tree arena was sized: {"bytes", "blocks_per_slot", "reclaim_half_blocks"} (the last 0 = never reclaimed)."""
        b, bps, half = C.c_uint64(), C.c_uint32(), C.c_uint32()
        check(self.L.c4_session_arena(self._h, C.byref(b), C.byref(bps), C.byref(half)))
        return {"bytes": b.value, "blocks_per_slot": bps.value, "reclaim_half_blocks": half.value}

    def start(self):
        check(self.L.c4_session_start(self._h))

    def step(self):
        check(self.L.c4_session_step(self._h))

    def set_timing(self, enable: bool):
        check(self.L.c4_session_set_timing(self._h, 1 if enable else 0))
        self._timing = bool(enable)

    def set_step_shape(self, games_per_wavefront: int):
        """Games per stepping wavefront of the fused output + step launch: 8 (default) or 4 (c4_session_set_step_shape).  A
        scheduling knob -- the records do not depend on it: 4 is 0.5 % faster beside a second session's kernels, 8 alone."""
        if not hasattr(self.L, "c4_session_set_step_shape"):   # only an older A/B library (C4A0_HIP_LIB): its one shape is 8
            return
        check(self.L.c4_session_set_step_shape(self._h, int(games_per_wavefront)))

    def round(self, evaluator: DeviceEvaluator):
        """One lock-step round: evaluate the leaves, step every game.  With an evaluator that offers its hidden activations
        (c4a0_amd.nn.InferenceNet) and the session in its default configuration with per-launch timing off (every HIP-graph
        capture), the heads' output layers run inside the step's launch (c4_session_step_head_out: one launch fewer on the
        round's chain, same bits); otherwise evaluate() then step()."""
        if (self.fuse_output_step and not self._timing and not self._extensions and getattr(self, "leaf_models", None) is None
                and getattr(evaluator, "fused_step_ok", False)):
            r = self.rows
            p, v = evaluator.forward_hidden(self.planes if r == self.n_slots else self.planes[:r])
            wp, wv, bp, bv = evaluator.head_out_operands()
            assert p.stride(1) == 1 and v.stride(1) == 1 and p.shape[0] == r
            check(self.L.c4_session_step_head_out(self._h, C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()), C.c_void_p(wp.data_ptr()),
                                                  C.c_void_p(wv.data_ptr()), C.c_void_p(bp.data_ptr()), C.c_void_p(bv.data_ptr()),
                                                  p.shape[1], p.stride(0), v.stride(0)))
            return
        if hasattr(evaluator, "round") and getattr(evaluator, "s", None) is self:   # the numpy-callback evaluator: answers handed over in the step's launch
            evaluator.round()
            return
        self.evaluate(evaluator)
        self.step()

    def capture_steps(self, evaluator: DeviceEvaluator, steps_per_graph: int = 8,
                      stream: Optional[torch.cuda.Stream] = None) -> "torch.cuda.CUDAGraph":
        """Capture `steps_per_graph` x (evaluator, step kernel) into one HIP graph.

        The evaluator must write into the session's bound tensors without host synchronisation
        (c4a0_amd.nn.InferenceNet does).  Per-launch device-clock timing is switched off (its
        sequence number would be frozen in the graph).  Replay with `graph.replay()` on `stream`
        (default: the current stream), to which the session stays bound.  Sessions that replay
        concurrently must each be captured on their own stream: library workspaces are per stream."""
        self.set_timing(False)
        main = stream if stream is not None else torch.cuda.current_stream(self.device)
        # warm the evaluator up outside the capture (library handles, autotuned kernels)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self.bind(side)
            for _ in range(2):
                self.evaluate(evaluator)
        main.wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with capture(graph, stream if stream is not None else torch.cuda.Stream(device=self.device)):   # (never the legacy default stream)
            self.bind(torch.cuda.current_stream(self.device))
            for _ in range(steps_per_graph):
                self.round(evaluator)
        self.bind(main)
        return graph

    # ---------------------------------------------------------------- results
    def counters(self) -> dict:
        c = Counters()
        check(self.L.c4_session_counters(self._h, C.byref(c)))
        return c.as_dict()

    def poll(self) -> Tuple[int, int]:
        done, err = C.c_uint64(), C.c_uint32()
        check(self.L.c4_session_poll(self._h, C.byref(done), C.byref(err)))
        return done.value, err.value

    def progress(self) -> Tuple[int, int, int]:
        """poll() plus the number of games taken off the request list: (done, started, error)."""
        done, started, err = C.c_uint64(), C.c_uint64(), C.c_uint32()
        check(self.L.c4_session_progress(self._h, C.byref(done), C.byref(started), C.byref(err)))
        return done.value, started.value, err.value

    def compact(self, multiple: int = 256) -> Tuple[int, int]:
        """Tail of a job (every request started): move the remaining games into the lowest slots and
        narrow the session to the smallest multiple of `multiple` slots holding them
        (c4_session_compact).  Returns (active games, rows); `self.rows` rows are evaluated and stepped
        from now on.  Graphs captured earlier are stale."""
        act, rows = C.c_uint32(), C.c_uint32()
        check(self.L.c4_session_compact(self._h, int(multiple), C.byref(act), C.byref(rows)))
        self.rows = rows.value
        return act.value, rows.value

    def raise_if_device_error(self):
        c = self.counters()
        if c["error"]:
            raise C4Error(c["error"], f"raised on device by slot {c['error_slot']}")

    def sample_counts(self) -> np.ndarray:
        out = np.zeros(max(1, self.n_games), dtype=np.uint32)
        check(self.L.c4_session_sample_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), self.n_games))
        return out[: self.n_games]

    def drain_samples(self) -> np.ndarray:
        """All samples of finished games, packed in reqs order, as a structured numpy array."""
        n = C.c_uint64()
        check(self.L.c4_session_drain_samples(self._h, None, 0, C.byref(n)))
        buf = np.zeros(max(1, n.value), dtype=SAMPLE_DTYPE)
        check(self.L.c4_session_drain_samples(self._h, buf.ctypes.data_as(C.POINTER(SampleRec)), n.value, C.byref(n)))
        return buf[: n.value]

    def pack_samples_device(self) -> torch.Tensor:
        """Finished games' records packed into one device tensor uint8[n, 64] (kernel K6)."""
        n = C.c_uint64()
        check(self.L.c4_session_pack_samples(self._h, None, 0, C.byref(n)))
        out = torch.empty((max(1, n.value), 64), dtype=torch.uint8, device=self.device)
        if n.value:
            check(self.L.c4_session_pack_samples(self._h, C.c_void_p(out.data_ptr()), n.value, C.byref(n)))
        return out[: n.value]

    def root_stats(self, slot: int = 0):
        pol = (C.c_float * 7)()
        qp, qn = C.c_float(), C.c_float()
        n, m, v = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self.L.c4_session_root_stats(self._h, slot, pol, C.byref(qp), C.byref(qn), C.byref(n), C.byref(m), C.byref(v)))
        return np.array(pol[:], dtype=np.float32), qp.value, qn.value, n.value, (m.value, v.value)

    def leaves(self, with_ordinals: bool = False):
        m = np.zeros(self.n_slots, dtype=np.uint64)
        v = np.zeros(self.n_slots, dtype=np.uint64)
        s = np.zeros(self.n_slots, dtype=np.uint32)
        o = np.zeros(self.n_slots, dtype=np.uint32)
        check(self.L.c4_session_leaves(self._h, m.ctypes.data_as(C.POINTER(C.c_uint64)), v.ctypes.data_as(C.POINTER(C.c_uint64)),
                                       s.ctypes.data_as(C.POINTER(C.c_uint32)), o.ctypes.data_as(C.POINTER(C.c_uint32))))
        return (m, v, s, o) if with_ordinals else (m, v, s)

    # ---------------------------------------------------------------- the loop
    def evaluate(self, evaluator: DeviceEvaluator):
        r = self.rows
        planes, logprobs, q = (self.planes, self.logprobs, self.q) if r == self.n_slots else (self.planes[:r], self.logprobs[:r], self.q[:r])
        if getattr(evaluator, "graph_safe", False):   # writes the bound tensors in place
            evaluator(planes, out_logprobs=logprobs, out_q=q)
            return
        lp, q_out = evaluator(planes)
        if lp.data_ptr() != logprobs.data_ptr():
            logprobs.copy_(lp.reshape(r, 7))
        if q_out.data_ptr() != q.data_ptr():
            q.copy_(q_out.reshape(r, 2))

    fuse_output_step = True    # round(): the heads' output layers inside the step's launch where the evaluator allows it (False: A/B)
    NARROW_CHECK_ROUNDS = 64   # lock-step rounds between two looks at the tail (a stream synchronisation each)

    def wants_narrowing(self, multiple: int = 256, asynchronous: bool = False) -> bool:
        """Tail of a job: True when every request has been started and at most half of the rows still
        hold a game, i.e. compact() would pay.  Takes no action.

        asynchronous=False: the decision is taken from a SYNCHRONOUS read of the session's counters, and
        callers ask at fixed round counts (NARROW_CHECK_ROUNDS), so the step at which a session narrows
        is a function of the games alone, not of host or GPU timing: with an evaluator whose low bits
        depend on the batch shape (a library GEMM) the job stays reproducible run to run.
        asynchronous=True (evaluators that declare `batch_invariant`, e.g. InferenceNet with the hand-written
        GEMMs): the samples do not depend on when the session narrows, so the decision reads the pinned
        progress probe instead -- no stream synchronisation, no drained run-ahead."""
        if getattr(self, "leaf_models", None) is not None or self.rows <= multiple:
            return False
        if asynchronous:
            done, started, _err = self.progress()   # as of an earlier step: at worst the narrowing comes a few rounds later
        else:
            c = self.counters()                     # synchronises this session's stream
            done, started = c["games_done"], c["games_started"]
        return started >= self.n_games and (self.n_games - done) <= self.rows // 2

    def narrow_if_worthwhile(self, multiple: int = 256, asynchronous: bool = False) -> bool:
        """wants_narrowing() -> compact().  True if `self.rows` changed (captured graphs are then stale).
        Only for a session whose kernels all run on ITS bound stream (compact() waits for that stream alone);
        sessions replayed from a shared graph on another stream go through session._run_pair."""
        if not self.wants_narrowing(multiple, asynchronous):
            return False
        before = self.rows
        self.compact(multiple)
        return self.rows != before

    def run(self, evaluator: DeviceEvaluator, max_steps: Optional[int] = None, poll_every: int = 16,
            on_step: Optional[Callable[[int], None]] = None, steps_per_graph: int = 0, phases: Optional[dict] = None,
            tail_steps_per_graph: int = 0) -> int:
        """Play all games set by set_games() to completion.  Returns the number of steps.

        steps_per_graph > 0 replays a HIP graph of that many (evaluator, step) rounds per host
        iteration -- for evaluators that are pure device code (no host callbacks); from the first narrowing of the tail on
        tail_steps_per_graph rounds (0 = the same).  phases (a dict) receives where the wall time went, as session._run_pair fills it."""
        self.bind()
        self.start()
        steps = 0
        t_c = time.perf_counter()
        graph = self.capture_steps(evaluator, steps_per_graph) if steps_per_graph > 0 else None
        if phases is not None:
            phases.update(first_capture_s=time.perf_counter() - t_c, recapture_s=0.0, captures=1 if graph is not None else 0, t_loop0=time.perf_counter())
        # The host enqueues much faster than the GPU executes.  Unthrottled it would run thousands of
        # steps ahead of the completion probe and the job would keep evaluating idle slots long after
        # the last game ended, so at most `max_chunks_in_flight` chunks (a graph replay, or
        # `poll_every` eager steps) are outstanding at any time.
        max_chunks_in_flight = 2
        inflight = []
        chunks = 0
        while True:
            if graph is not None:
                graph.replay()
                steps += steps_per_graph
                chunk_end = True
            else:
                if on_step is not None:   # observers see the evaluator outputs before the step consumes them
                    self.evaluate(evaluator)
                    on_step(steps)
                    self.step()
                else:
                    self.round(evaluator)
                steps += 1
                chunk_end = steps % poll_every == 0
            if chunk_end:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                inflight.append(ev)
                if len(inflight) > max_chunks_in_flight:
                    inflight.pop(0).synchronize()
                done, started, err = self.progress()
                if err:
                    self.raise_if_device_error()
                if phases is not None and started >= self.n_games and "t_all_started" not in phases:
                    phases["t_all_started"], phases["steps_all_started"] = time.perf_counter(), steps   # as the (slightly late) probe sees it
                if done >= self.n_games:
                    break
                chunks += 1
                if graph is not None and chunks % max(1, self.NARROW_CHECK_ROUNDS // steps_per_graph) == 0 and \
                        self.narrow_if_worthwhile(asynchronous=bool(getattr(evaluator, "batch_invariant", False))):
                    inflight.clear()
                    t_c = time.perf_counter()
                    steps_per_graph = int(tail_steps_per_graph or steps_per_graph)
                    graph = self.capture_steps(evaluator, steps_per_graph)   # the old graph carries the old width
                    if phases is not None:
                        phases["recapture_s"] += time.perf_counter() - t_c
                        phases["captures"] += 1
                        phases.setdefault("narrowings", []).append((steps, [self.rows]))
            if max_steps is not None and steps >= max_steps:
                break
        if graph is not None:
            torch.cuda.synchronize(self.device)
            self.set_timing(True)
        c = self.counters()
        if c["error"]:
            raise C4Error(c["error"], f"raised on device by slot {c['error_slot']}")
        return steps


PAIRED_STEP_GAMES_PER_WAVEFRONT = 4   # (8: the A/B)


def capture_pair(sessions: Sequence["DeviceSession"], streams: Sequence[torch.cuda.Stream], evaluator, steps_per_graph: int = 32,
                 strict: bool = False, offset_stage: int = 1) -> "torch.cuda.CUDAGraph":
    """TWO sessions' rounds captured into ONE HIP graph with an explicit software pipeline between them.

    Two sessions that replay independent graphs on two streams settle into whatever relative phase the
    contention between them happens to lock: measured on MI355X the same binary runs at 22.5 k or 25.7 k
    games/s (and a faster tower moved it to 23.1 k or 27.5 k) depending on that phase -- in phase, both
    sessions' latency-bound kernels (head outputs, step kernel) leave the chip idle together.  Here the
    phase is part of the graph: in the first round of every replay session B's [tower, first hidden layer]
    starts only when session A's has finished (a cross-stream event recorded and awaited during capture: an
    edge of the graph), i.e. B runs half a round behind A, each session's heavy half beside the OTHER's
    [narrow layers, head outputs, step kernel]; for the rest of the replay the two streams run free from
    that phase, and the join at the graph's end re-aligns them.  Nothing else changes (same kernels, same
    per-session order, same samples).  Measured on one box (profiles/r03_pairing.txt): this 25.7 k games/s
    every time; a hand-over in EVERY round (strict=True) 25.4-25.5 k; starting B after A's third GEMM (offset_stage=4)
    instead 22.8 k (the bad phase, reproduced on purpose); two free-running graphs 26.1-26.2 k in four runs
    out of five and 24.3 k in the fifth (22.7 k on other boxes).  What the single graph buys is that the
    result no longer depends on a coin toss at start-up, for 1.6 % of the lucky case.

    `evaluator` must be a c4a0_amd.nn.InferenceNet (its `stage_hook` marks the two points).  The graph
    is replayed on streams[0]; both sessions stay bound to their streams."""
    a, b = sessions
    s0, s1 = streams
    dev = a.device
    for s in sessions:
        s.set_timing(False)
        s.set_step_shape(PAIRED_STEP_GAMES_PER_WAVEFRONT)   # (profiles/r05_out_step_gpw.txt: +0.5 % beside the other session's kernels)
    # warm the evaluator up outside the capture, once per stream (library handles, lazy module loads, the LDS opt-in of a tile
    # shape first used at this width); evaluating the current leaves once more changes nothing a game sees
    for s, st in zip(sessions, streams):
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(st)
        with torch.cuda.stream(side):
            s.bind(side)
            for _ in range(2):
                s.evaluate(evaluator)
        st.wait_stream(side)
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    prev_hook = getattr(evaluator, "stage_hook", None)
    try:
        with capture(graph, s0):
            a.bind(s0)
            b.bind(s1)
            s1.wait_stream(s0)                                  # fork: s1 joins the capture
            ev_b_prev = None
            # strict: hand-over in EVERY round instead of once per graph; offset_stage: B starts when this stage of A's
            # first round is done (1 = first hidden layer) -- measurement arguments (tools/), the defaults are the product
            for r in range(steps_per_graph):
                ev_a, ev_b = torch.cuda.Event(), torch.cuda.Event()
                free = not strict

                def hook_a(stage, ev_a=ev_a, wait_for=ev_b_prev, free=free, r=r):
                    if free:
                        if r == 0 and stage == offset_stage:
                            ev_a.record(s0)
                        return
                    if stage == 0 and wait_for is not None:
                        s0.wait_event(wait_for)                 # A's heavy half after B's previous one
                    elif stage == 1:
                        ev_a.record(s0)

                def hook_b(stage, ev_a=ev_a, ev_b=ev_b, free=free, r=r):
                    if free:
                        if r == 0 and stage == 0:
                            s1.wait_event(ev_a)
                        return
                    if stage == 0:
                        s1.wait_event(ev_a)                     # B's heavy half after A's
                    elif stage == 1:
                        ev_b.record(s1)

                evaluator.stage_hook = hook_a
                with torch.cuda.stream(s0):
                    a.round(evaluator)
                evaluator.stage_hook = hook_b
                with torch.cuda.stream(s1):
                    b.round(evaluator)
                ev_b_prev = ev_b
            s0.wait_stream(s1)                                  # join
    finally:
        evaluator.stage_hook = prev_hook
    return graph


def run_sessions(sessions: Sequence["DeviceSession"], evaluator: DeviceEvaluator, steps_per_graph: int = 8,
                 max_chunks_in_flight: int = 2, paired: Optional[bool] = None, phases: Optional[dict] = None,
                 tail_steps_per_graph: Optional[int] = None) -> List[int]:
    """Play the games of several sessions of one device to completion CONCURRENTLY, each session
    replaying its own HIP graph of (evaluator, step kernel) rounds on its own stream.

    Why: one step of one session is a strict chain evaluator -> step kernel, and several links of it
    (the step kernel, the heads' output kernel, the tower's prologue) are latency-bound and leave
    most of the chip idle.  With the resident games split over two sessions, one half's latency-bound
    kernels run under the other half's GEMMs.  A game's samples do not depend on which session or
    slot plays it.  `evaluator` must be pure device code writing into the bound tensors
    (graph_safe, e.g. c4a0_amd.nn.InferenceNet); it is shared (weights are read-only, activations
    live in each graph's own pool).  paired: None = ONE explicitly pipelined graph (capture_pair) whenever two sessions
    share an InferenceNet; False = independent graphs (A/B).  Returns the steps each session ran."""
    dev = sessions[0].device
    streams = [torch.cuda.Stream(device=dev) for _ in sessions]
    cur = torch.cuda.current_stream(dev)
    invariant = bool(getattr(evaluator, "batch_invariant", False))
    paired = len(sessions) == 2 and hasattr(evaluator, "stage_hook") and paired is not False
    graphs = []
    for s, st in zip(sessions, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            s.bind(st)
            s.start()
        st.synchronize()
        if not paired:
            graphs.append(s.capture_steps(evaluator, steps_per_graph, stream=st))
    if paired:
        return _run_pair(sessions, streams, evaluator, steps_per_graph, max_chunks_in_flight, invariant, phases, tail_steps_per_graph)
    steps = [0] * len(sessions)
    chunks = [0] * len(sessions)
    live = [s.n_games > 0 for s in sessions]
    inflight: List[List[torch.cuda.Event]] = [[] for _ in sessions]
    while any(live):
        for i, (s, st, g) in enumerate(zip(sessions, streams, graphs)):
            if not live[i]:
                continue
            with torch.cuda.stream(st):
                g.replay()
                ev = torch.cuda.Event()
                ev.record(st)
                steps[i] += steps_per_graph
                inflight[i].append(ev)
                if len(inflight[i]) > max_chunks_in_flight:   # bounded host run-ahead, as in DeviceSession.run
                    inflight[i].pop(0).synchronize()
                done, err = s.poll()
            if err:
                torch.cuda.synchronize(dev)
                s.raise_if_device_error()
            chunks[i] += 1
            if done >= s.n_games:
                live[i] = False
            elif chunks[i] % max(1, s.NARROW_CHECK_ROUNDS // steps_per_graph) == 0 and s.narrow_if_worthwhile(asynchronous=invariant):   # tail: fewer rows to evaluate, see DeviceSession.compact
                inflight[i].clear()
                graphs[i] = s.capture_steps(evaluator, steps_per_graph, stream=st)
    torch.cuda.synchronize(dev)
    for s in sessions:
        s.set_timing(True)
        s.raise_if_device_error()
    return steps


def _run_pair(sessions, streams, evaluator, steps_per_graph, max_chunks_in_flight, invariant, phases=None, tail_steps_per_graph=None) -> List[int]:
    """run_sessions for two sessions of an InferenceNet: ONE graph that pipelines the two explicitly
    (capture_pair), replayed until both sessions' games are over.  A session that finishes first keeps
    stepping its idle slots until the other is done (its rows are narrowed away like any tail).

    steps_per_graph rounds per replay while slots are refilled (long graphs amortise the replay boundary: at BASELINE
    config 2, 8 -> 0.109 ms per round, 64 -> 0.104); from the first narrowing on -- the graph has to be captured again there
    anyway -- tail_steps_per_graph rounds (default: the same), so that the job's end is noticed within a few short replays."""
    dev = sessions[0].device
    t_c = time.perf_counter()
    graph = capture_pair(sessions, streams, evaluator, steps_per_graph)
    if phases is not None:
        phases.update(first_capture_s=time.perf_counter() - t_c, recapture_s=0.0, captures=1, t_loop0=time.perf_counter())
    per_graph = steps_per_graph
    tail_steps = int(tail_steps_per_graph or steps_per_graph)
    steps = 0
    since_check = 0
    inflight: List[torch.cuda.Event] = []
    while True:
        with torch.cuda.stream(streams[0]):
            graph.replay()
            ev = torch.cuda.Event()
            ev.record(streams[0])
        steps += per_graph
        since_check += per_graph
        inflight.append(ev)
        if len(inflight) > max_chunks_in_flight:   # bounded host run-ahead, as in DeviceSession.run
            inflight.pop(0).synchronize()
        done_all, started_all = True, True
        for s, st in zip(sessions, streams):
            with torch.cuda.stream(st):
                done, started, err = s.progress()
            if err:
                torch.cuda.synchronize(dev)
                s.raise_if_device_error()
            done_all = done_all and done >= s.n_games
            started_all = started_all and started >= s.n_games
        if phases is not None and started_all and "t_all_started" not in phases:
            phases["t_all_started"], phases["steps_all_started"] = time.perf_counter(), steps   # as the (slightly late) probe sees it
        if done_all:
            break
        if since_check >= sessions[0].NARROW_CHECK_ROUNDS and started_all:
            since_check = 0
            # Both sessions' kernels are nodes of ONE graph replayed on streams[0], and up to max_chunks_in_flight
            # replays are still running here; compact() waits for its session's OWN stream only (B's is streams[1],
            # on which nothing of the replay is visible).  So: decide first, for both, without touching the slots;
            # if either wants to narrow, drain the device; only then move games (compacting B under a
            # running replay corrupted its slots whenever B narrowed at a check where A did not).
            if not invariant:
                torch.cuda.synchronize(dev)          # the synchronous decision below must see every replayed round
            wants = [s.wants_narrowing(asynchronous=invariant) for s in sessions]
            if any(wants):
                inflight.clear()
                torch.cuda.synchronize(dev)
                before = [s.rows for s in sessions]
                for s, w in zip(sessions, wants):
                    if w:
                        s.compact()
                if [s.rows for s in sessions] != before:   # the old graph carries the old widths
                    t_c = time.perf_counter()
                    per_graph = tail_steps
                    graph = capture_pair(sessions, streams, evaluator, per_graph)
                    if phases is not None:
                        phases["recapture_s"] += time.perf_counter() - t_c
                        phases["captures"] += 1
                        phases.setdefault("narrowings", []).append((steps, [s.rows for s in sessions]))
    torch.cuda.synchronize(dev)
    for s in sessions:
        s.set_timing(True)
        s.raise_if_device_error()
    return [steps, steps]


SAMPLE_DTYPE = np.dtype([("game_id", "<u8"), ("mask", "<u8"), ("value", "<u8"), ("policy", "<f4", (7,)),
                         ("q_penalty", "<f4"), ("q_no_penalty", "<f4"), ("meta", "<u4")])
assert SAMPLE_DTYPE.itemsize == 64
