#!/usr/bin/env python3
"""bench.py -- self-play games/sec of the MI355X-native generator (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One lock-step ROUND = one MCTS simulation for every resident game: the ResNet evaluates the G leaf
positions (HIP-graph replay, bf16) and the fused HIP step kernel consumes the outputs (expand,
backup, move/finish/refill, select, encode the next leaves; a game whose new leaf is terminal
runs that simulation in the same launch).  The G games are split over two sessions whose graphs
replay concurrently on two streams (--sessions); a round advances both.  One bench "step" = R
rounds (--rounds-per-step, R = 704 = 11 replays of the 64-round graph; `config.rounds_per_step`):
the hot path over one batch of work large enough that `--steps 20` completes >= 10 x 4 096 games
inside the timed region.  Workload at every N:
BASELINE config 2 per GPU -- 4 096 concurrent games, n_mcts_iterations = 100, 4-block/32-channel
ResNet in bf16, c_exploration 6.6, c_ply_penalty 0.01, game ids sharded id % N (weak scaling;
config 3 is exactly this at N = 8).  Synthetic data: empty-board starts, random-init network
(torch.manual_seed(1337)).  Before the warm-up the session is rolled forward until finished
games have been replaced at least once, so the timed steps see the steady-state mix of game
phases; `value` counts games COMPLETED inside the timed steps.

Prints ONE JSON line (rank 0).  Extra objects: "roofline" (fused tree step kernel vs HBM, always
>= 300 event-bracketed launches), "cpu_baseline" (the CPU oracle in the reference's thread topology
-- one evaluator thread + workers, two queues -- timed on this box's host cores), "nn" (evaluator
FLOP rate), and for N > 1 "sample_allgather" (the path's one exchange step, merged and checked).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


def algorithmic_bytes(c: dict, planes_bytes_per_elem: int) -> dict:
    """SURVEY 8(d) canonical-node traffic model, from the DEVICE-counted S, K, E:
    select/backup 92*S + 28*K, expand 332*E, leaf encode 84 elements per sim."""
    sb = 92 * c["select_levels"] + 28 * c["backup_nodes"]
    ex = 332 * c["expansions"]
    enc = 84 * planes_bytes_per_elem * c["sims"]
    return {"select_backup": sb, "expand": ex, "encode": enc, "total": sb + ex + enc}


def step_kernel_source_hash() -> str:
    """sha256 over the two files the step kernel is compiled from (profiles/step_kernel_traffic.json records it)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("c4_session.hip", "c4_device.hpp"):
        with open(os.path.join(ROOT, "c4a0_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def kernel_timeline(sess, stream, net, n_rounds: int = 300, batch: int = 100):
    """Per-kernel launch durations of ONE session's round as the timed region launches it -- tower, the heads' hidden-layer GEMMs,
    and the fused output + step kernel (DeviceSession.round with per-launch timing off) -- from HIP events recorded on the session's
    stream at every launch boundary (InferenceNet.stage_hook marks them).  The rounds are queued behind a blocker (a few large
    matmuls) in batches, so the host has enqueued a whole batch before the GPU reaches it and the kernels run back to back; the
    other session is idle, which is also what rocprofv3's kernel trace measures (it serialises the queues).
    Returns ([(label, total_ms, launches)] in launch order, rounds measured, the interval of a one-element kernel in us)."""
    sess.set_timing(False)
    blocker = torch.ones((8192, 8192), dtype=torch.bfloat16, device=sess.device)
    totals, order = {}, []
    prev_hook = net.stage_hook
    done = 0
    try:
        while done < n_rounds:
            evs = []
            with torch.cuda.stream(stream):
                for _ in range(16):
                    torch.mm(blocker, blocker)              # ~2 ms each: the batch below is fully queued before it starts
                for _ in range(batch):
                    marks = []

                    def hook(stage, marks=marks):
                        e = torch.cuda.Event(enable_timing=True)
                        e.record(stream)
                        marks.append((stage, e))
                    net.stage_hook = hook
                    sess.round(net)
                    end = torch.cuda.Event(enable_timing=True)
                    end.record(stream)
                    evs.append((marks, end))
            stream.synchronize()
            for marks, end in evs:
                seq = marks + [(-1, end)]
                for i in range(len(seq) - 1):
                    # hook(0) precedes the tower's launch, hook(2) follows it, hook(1) follows the first hidden layer's, hook(3 + i) the
                    # policy head's further layers'; what follows the last mark is the fused output + step launch (nn.py forward_hidden)
                    st_a = seq[i][0]
                    label = "tower" if st_a == 0 else "gemm_first_hidden" if st_a == 2 else "out_step" if i == len(seq) - 2 else "gemm_narrow"
                    if label not in totals:
                        totals[label] = [0.0, 0]
                        order.append(label)
                    totals[label][0] += seq[i][1].elapsed_time(seq[i + 1][1])
                    totals[label][1] += 1
            done += batch
        # what an event-to-event interval holds besides the kernel: the same bracket around a chain of one-element kernels (c4_expf_logf,
        # ~1 us of work each): dispatch of a dependent launch + the shortest kernel there is
        import ctypes as C
        from c4a0_amd import _lib
        x = torch.ones(1, dtype=torch.float32, device=sess.device)
        y = torch.empty_like(x)
        n_null = 400
        with torch.cuda.stream(stream):
            for _ in range(8):
                torch.mm(blocker, blocker)
            a_ev, b_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a_ev.record(stream)
            for _ in range(n_null):
                _lib.check(_lib.lib().c4_expf_logf(C.c_void_p(x.data_ptr()), 1, 0, C.c_void_p(y.data_ptr()), C.c_void_p(stream.cuda_stream)))
            b_ev.record(stream)
        stream.synchronize()
        null_us = a_ev.elapsed_time(b_ev) * 1e3 / n_null
    finally:
        net.stage_hook = prev_hook
    return [(k, totals[k][0], totals[k][1]) for k in order], done, null_us


def usable_cores() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(net, device, n_iter: int, threads: int, budget_s: float = 20.0):
    """The reference's CPU path restated IN ITS OWN TOPOLOGY (self_play.rs:60-106): this thread is the
    NN thread (NNThread::loop_until_close: drain the queue, batch the unique leaves, call the
    evaluator), `threads - 1` C worker threads are the MctsThreads, games travel over two queues, so
    network evaluation and tree work overlap (oracle c4o_self_play_async).  The evaluator is the SAME
    bf16 network on the GPU through the numpy callback round trip of nn.py:119-130 (InferenceNet.forward_numpy).

    Reported as the MEDIAN of >= 3 equal samples of the same workload (with min / max / n): a probe
    sizes the sample so that the three together take about `budget_s` seconds.  The NN thread keeps
    one core to itself and the workers are pinned one per core over the others
    (c4o_set_thread_pinning): 15 spinning workers preempting the thread that feeds them is what made
    the single-sample figure of round 2 move by +-25 % run to run."""
    from oracle import c4oracle as O

    cb_time = [0.0]

    def cb(_model_id, x):   # the reference caller's callback (training.py:179-189): model.forward_numpy(x)
        t = time.perf_counter()
        out = net.forward_numpy(x)
        cb_time[0] += time.perf_counter() - t
        return out

    threads = max(2, threads)
    O.lib().c4o_set_thread_pinning(1)

    def run(n_games, topology="async"):
        reqs = [(i, 0, 0) for i in range(n_games)]
        cb_time[0] = 0.0
        t0 = time.perf_counter()
        _res, st = O.self_play(reqs, 4096, n_iter, 6.6, 0.01, cb, n_threads=threads, topology=topology)
        dt = time.perf_counter() - t0
        return dt, st, cb_time[0] / dt

    # size the sample: all games resident at once, as in the reference, so the evaluator batches grow with it
    n_samples = 3
    n_games, (dt, st, share) = 256, run(256)
    while dt < budget_s / (2 * n_samples) and n_games < 32768:
        n_games *= 2 if dt > budget_s / (8 * n_samples) else 4
        dt, st, share = run(n_games)
    runs = [(dt, st, share)] + [run(n_games) for _ in range(n_samples - 1)]
    rates = sorted(n_games / r[0] for r in runs)
    med_i = sorted(range(len(runs)), key=lambda i: runs[i][0])[len(runs) // 2]
    dt, st, share = runs[med_i]
    # context 1: the round-1 restatement (lock-step ticks: evaluation and tree work serialised) on a quarter of the sample
    dt_l, _st_l, _ = run(max(256, n_games // 4), topology="lockstep")
    # context 2: the tree path alone on the host cores (uniform evaluator, no network at all)
    t0 = time.perf_counter()
    _r, st_u = O.self_play([(i, 0, 0) for i in range(2048)], 4096, n_iter, 6.6, 0.01, "uniform", n_threads=threads, topology="async")
    tree_only = st_u["sims"] / (time.perf_counter() - t0)
    O.lib().c4o_set_thread_pinning(0)
    return {"value": rates[len(rates) // 2], "min": rates[0], "max": rates[-1], "n": len(rates), "unit": "games/s", "cores": threads, "kind": "port",
            "topology": f"async: 1 evaluator thread (own core) + {threads - 1} MCTS worker threads pinned over the other cores, two queues, "
                        "evaluation and tree work overlapped (self_play.rs:60-106)",
            "evaluator_callback_share_of_wall": share,
            "tree_only_sims_per_s": tree_only, "lockstep_games_per_s": max(256, n_games // 4) / dt_l,
            "nn_calls": st["nn_calls"], "mean_nn_batch": st["nn_positions"] / max(1, st["nn_calls"]),
            "sample": f"median of {len(rates)} runs of {n_games} games each, n_mcts_iterations={n_iter}, C oracle in the reference's thread topology "
                      f"({threads} threads) + the same bf16 ResNet on the GPU via the numpy callback round trip; {dt:.1f} s per run",
            "sims_per_s": st["sims"] / dt}


def whole_job(args, device, real_stdout):
    """`--whole-job`: the reference's OWN default self-play job (src/c4a0/main.py:40-51: 1 700 games,
    n_mcts_iterations = 1 400, batch 2 000, 1-block / 32-channel network with 4 policy and 2 value
    layers), played start to finish through `play_games` -- session set-up, HIP-graph capture, the
    tail where finished slots idle, and the sample hand-over included (after one untimed 64-game job per
    mode: process start-up is not job time) -- in the three ways a caller
    can use it: the unmodified numpy callback (training.py:179-189), the same call with a
    `DeviceCallback` wrapper, and `evaluator=`.  Not the headline: one JSON line of its own."""
    import pickle

    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(1, 32, 4, 2)), device, dtype=torch.bfloat16)
    n_games, n_iter = args.whole_job_games, args.whole_job_n_mcts
    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(n_games)]

    def cb(_model_id, x):   # exactly the reference caller's callback (training.py:179-189): lambda model_id, x: model.forward_numpy(x)
        return net.forward_numpy(x)

    out, ref = {}, None
    modes = (("device_mode", dict(evaluator=net)),
             # A/B row: the never-reclaimed arena of rounds 1-4 (13 GB for this job) -- what reclaiming the arenas during play costs
             ("device_mode_never_reclaimed_arena", dict(evaluator=net, reclaim=False)),
             ("device_callback_wrapper", dict(py_eval_pos_cb=c4a0_amd.DeviceCallback(net, device))),
             ("numpy_callback", dict(py_eval_pos_cb=cb)),
             # EXTENSION rows (not the reference's algorithm, off by default): the evaluation cache answers a leaf
             # whose position the evaluator has already seen without an evaluator row -- same samples, because
             # the evaluator is a function of the position alone -- fewer lock-step rounds
             ("extension_eval_cache_device_mode", dict(evaluator=net, eval_cache_entries=1 << 24)),
             ("extension_eval_cache_numpy_callback", dict(py_eval_pos_cb=cb, eval_cache_entries=1 << 24)))
    wanted = [m for m in args.whole_job_modes.split(",") if m]
    for name, kw in modes:
        if wanted and name not in wanted:
            continue
        st = {}
        # untimed: a 64-game job the same way first, so that no mode's figure carries the process's one-time costs
        # (code-object loading, the LDS opt-in, allocator warm-up) -- a training loop calls play_games every generation
        c4a0_amd.play_games(reqs[:64], 2000, 20, 6.6, 0.01, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = c4a0_amd.play_games(reqs, 2000, n_iter, 6.6, 0.01, stats=st, **kw)
        recs, _counts = res.to_records()
        dt = time.perf_counter() - t0
        # what the reference's generation does next (src/c4a0/training.py:62-63): pickle.dump(games, f) = __getstate__ = to_cbor
        t0 = time.perf_counter()
        blob = pickle.dumps(res)
        dt_dump = time.perf_counter() - t0
        t0 = time.perf_counter()
        back = pickle.loads(blob)
        dt_load = time.perf_counter() - t0
        ref = recs if ref is None else ref
        out[name] = {"games_per_s": n_games / dt, "sims_per_s": st["sims"] / dt, "seconds": dt, "steps": st["steps"],
                     "samples": int(len(recs)), "samples_identical_to_device_mode": bool(recs.tobytes() == ref.tobytes()),
                     "pickle_seconds": dt_dump, "unpickle_seconds": dt_load, "pickle_bytes": len(blob),
                     "pickle_round_trip_identical": bool(back.to_records()[0].tobytes() == recs.tobytes()),
                     "games_per_s_play_plus_pickle": n_games / (dt + dt_dump),
                     "phases_s": {k: st["phases"][k] for k in ("setup_s", "start_and_capture_s", "steady_s", "tail_s", "drain_s")},
                     "arena_reclaim_passes": st.get("reclaim_passes", 0), "arena_reclaim_blocks": st.get("reclaim_blocks", 0)}
        if "eval_cache_entries" in kw:
            out[name]["cache_hit_rate"] = st["eval_cache_hits"] / max(1, st["eval_cache_probes"])
    line = {"metric": "whole-job self-play games/sec, the reference's default job", "value": out["device_mode"]["games_per_s"], "unit": "games/s",
            "n_gpus": 1, "higher_is_better": True, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"src/c4a0/main.py:40-51 defaults: {n_games} games, n_mcts_iterations={n_iter}, max_nn_batch_size=2000, 1-block/32-ch ResNet (4 policy / 2 value layers) bf16"},
            **out}
    os.write(real_stdout, (json.dumps(line) + "\n").encode())


def product_legs(args, device, real_stdout):
    """`--product-legs`: the PRODUCT loop under the driver's clock: whole `c4a0_amd.play_games(reqs, ...,
    evaluator=net)` calls at BASELINE config 2's shape -- reqs in, PlayGamesResult out; session set-up, graph captures, completion
    polling, the tail where finished slots idle, narrowing, the sample hand-over and the pickling the reference does next all
    inside the clock -- where the headline replays captured graphs over an endless queue:
      whole_call_10xG          40 960 games (10 x G, SURVEY 8d's own N for config 2), resident games chosen by the library
      whole_call_10xG_4096     the same job on exactly 4 096 slots (config 2's "4 096 concurrent games") -- and its records must equal the first's
      one_generation_4096      4 096 games on 4 096 slots: what each rank of BASELINE config 3 really runs (no refill: the whole job is tail)
      python_host_loop_10xG    the same 40 960-game job driven by the Python loop of c4a0_amd/session.py (host_loop="python": what every other
                               kind of evaluator gets) instead of `c4_play_games_bf16`, the library's own host loop, which `play_games` takes
                               by itself for an InferenceNet (one native call: what a Rust host of the reference binds in place of
                               self_play()); the two must return the same records
      eval_cache_10xG          EXTENSION (off by default, not the reference's algorithm: it evaluates every leaf and dedups inside a
                               batch only, self_play.rs:203-208): leaves whose position the evaluator has already answered skip the
                               evaluator row; records must equal the cache-off job's -- all of them are compared
    each with the call's wall time split into set-up / capture / steady / tail / drain (stats["phases"])."""
    import hashlib
    import pickle

    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(ModelConfig(4, 32, 4, 2)), device, dtype=torch.bfloat16)
    G = 4096
    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(10 * G)]
    c4a0_amd.play_games(reqs[:64], 4096, 20, 6.6, 0.01, evaluator=net)      # untimed: code objects, LDS opt-ins, allocator warm-up
    c4a0_amd.play_games(reqs[:G], 4096, 10, 6.6, 0.01, evaluator=net)       # ... and the tree arena of the big shapes (kept by the library between sessions)

    def call(n, **kw):
        # every shape is played TWICE and the second call is the one reported (a training loop calls play_games every generation): the
        # first call of a shape also pays for what the process keeps afterwards -- the tree arena, PyTorch's device and pinned blocks for
        # the hand-over (a 48 MB transfer into fresh pageable memory: 0.05-0.2 s once) -- and is reported beside it as first_call_seconds
        first = None
        for _rep in range(2):
            st = {}
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = c4a0_amd.play_games(reqs[:n], 4096, 100, 6.6, 0.01, evaluator=net, stats=st, **kw)
            dt = time.perf_counter() - t0
            first = dt if first is None else first
        t0 = time.perf_counter()
        blob = pickle.dumps(res)
        dt_dump = time.perf_counter() - t0
        recs, counts = res.to_records()
        ph = st["phases"]
        return recs, {"games": n, "host_loop": st["host_loop"], "sessions": st["concurrent_sessions"], "resident_games": st["n_slots"], "seconds": dt, "first_call_seconds": first, "games_per_s": n / dt, "sims_per_s": st["sims"] / dt, "rounds": st["steps"],
                      "samples": int(len(recs)), "phases_s": {k: ph[k] for k in ("setup_s", "start_and_capture_s", "steady_s", "tail_s", "drain_s")},
                      "tail_share_of_call": ph["tail_s"] / dt, "graph_captures": ph["graph_captures"], "rounds_until_all_started": ph["rounds_until_all_started"],
                      "pickle_seconds": dt_dump, "pickle_bytes": len(blob), "games_per_s_play_plus_pickle": n / (dt + dt_dump),
                      "records_sha256": hashlib.sha256(recs.tobytes()).hexdigest()[:16],
                      **({"cache_hit_rate": st["eval_cache_hits"] / max(1, st["eval_cache_probes"])} if kw.get("eval_cache_entries") else {})}

    out = {}
    base, out["whole_call_10xG"] = call(10 * G)
    again, out["whole_call_10xG_4096"] = call(10 * G, resident_games=G)
    out["whole_call_10xG_4096"]["records_identical_to_whole_call_10xG"] = bool(again.tobytes() == base.tobytes())
    _r, out["one_generation_4096"] = call(G, resident_games=G)
    py, out["python_host_loop_10xG"] = call(10 * G, host_loop="python")
    out["python_host_loop_10xG"]["records_identical_to_whole_call_10xG"] = bool(py.tobytes() == base.tobytes())
    out["whole_call_10xG"]["entry_point"] = "c4_play_games_bf16 (include/c4a0_hip.h): sessions, graphs, polling, narrowing and the merged hand-over inside one C call"
    cached, out["eval_cache_10xG"] = call(10 * G, eval_cache_entries=1 << 24)
    out["eval_cache_10xG"]["samples_identical"] = bool(cached.tobytes() == base.tobytes())
    out["eval_cache_10xG"]["samples_compared"] = int(len(base))
    out["eval_cache_10xG"]["note"] = "EXTENSION, off by default; not part of the headline"
    os.write(real_stdout, (json.dumps(out) + "\n").encode())


def config1_leg(device):
    """BASELINE config 1 ("for C1, the CPU ResNet through the numpy callback"): 32 self-play games, n_mcts_iterations = 10,
    a random-init 1-block / 32-channel ResNet -- the reference's own CPU-runnable plumbing case.  Two figures for the same job:
    (a) the reference's CPU path restated: the C oracle in the reference's thread topology (self_play.rs:60-106) with the network
        evaluated ON THE HOST CORES in f32 by PyTorch through the numpy callback, exactly as `forward_numpy` does it
        (src/c4a0/nn.py:119-130 minus the device copies);
    (b) this library: `play_games` with the same network on the GPU (bf16, hand-written kernels), whole call.
    The job is tiny (32 games, ~15 moves of <= 10 simulations each): both figures are latency, not throughput."""
    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
    from oracle import c4oracle as O

    torch.manual_seed(1337)
    model = ConnectFourNet(ModelConfig(1, 32, 4, 2)).eval()
    # ONE intra-op thread: the evaluator thread has a core to itself and the oracle's workers SPIN on theirs (pinned); a PyTorch
    # thread pool scheduled onto those cores crawls (first version of this leg: minutes instead of seconds on the 16-core box)
    torch.set_num_threads(1)

    @torch.no_grad()
    def cpu_forward_numpy(_model_id, x):      # nn.py:119-130 on the host
        lp, qp, qn = model(torch.from_numpy(x))
        return np.ascontiguousarray(lp.numpy()), np.ascontiguousarray(qp.numpy()), np.ascontiguousarray(qn.numpy())

    reqs = [(i, 0, 0) for i in range(32)]
    threads = max(2, min(usable_cores(), 64))
    O.lib().c4o_set_thread_pinning(1)
    cpu_s, st = [], None
    for _ in range(5):
        t0 = time.perf_counter()
        _res, st = O.self_play(reqs, 2000, 10, 6.6, 0.01, cpu_forward_numpy, n_threads=threads, topology="async")
        cpu_s.append(time.perf_counter() - t0)
    O.lib().c4o_set_thread_pinning(0)
    cpu_s.sort()
    net = InferenceNet(model, device, dtype=torch.bfloat16)
    metas = [c4a0_amd.GameMetadata(*r) for r in reqs]
    c4a0_amd.play_games(metas, 2000, 10, 6.6, 0.01, evaluator=net, device=device)      # untimed: code objects, LDS opt-ins
    gpu_s = []
    stats = {}
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = c4a0_amd.play_games(metas, 2000, 10, 6.6, 0.01, evaluator=net, device=device, stats=stats)
        gpu_s.append(time.perf_counter() - t0)
    gpu_s.sort()
    return {"workload": "BASELINE config 1: 32 self-play games, n_mcts_iterations=10, random-init 1-block/32-ch ResNet (4 policy / 2 value layers)",
            "cpu_reference_path": {"games_per_s": 32 / cpu_s[2], "seconds_median_of_5": cpu_s[2], "cores": threads, "kind": "port",
                                   "evaluator": f"f32 PyTorch on the host ({torch.get_num_threads()} threads) through the numpy callback",
                                   "nn_calls": st["nn_calls"], "mean_nn_batch": st["nn_positions"] / max(1, st["nn_calls"]), "sims": st["sims"]},
            "hip": {"games_per_s": 32 / gpu_s[2], "seconds_median_of_5": gpu_s[2], "sims": stats["sims"], "steps": stats["steps"],
                    "samples": int(sum(len(r.samples) for r in res.results)), "evaluator": "bf16, hand-written HIP kernels, device mode (whole play_games call)"}}


def run_child(cmd, timeout_s: float):
    """A time-limited child process that can never hold the headline hostage: output goes to temporary FILES
    (no pipe to drain), the child is polled against a deadline, and a child that does not die after SIGKILL -- e.g. one
    stuck in an uninterruptible GPU wait -- is abandoned after 10 more seconds instead of waited for.  (subprocess.run's
    own timeout kills and then waits WITHOUT a limit.)  Returns (returncode or None, stdout, stderr tail)."""
    import subprocess
    import tempfile

    with tempfile.TemporaryFile() as fo, tempfile.TemporaryFile() as fe:
        p = subprocess.Popen(cmd, stdout=fo, stderr=fe, cwd=ROOT)
        deadline = time.monotonic() + timeout_s
        rc = None
        while rc is None and time.monotonic() < deadline:
            try:
                rc = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                pass
        timed_out = rc is None
        if timed_out:
            p.kill()                      # exactly this pid
            try:
                p.wait(timeout=10.0)
            except subprocess.TimeoutExpired:
                pass                      # abandoned: the line below is printed regardless
        fo.seek(0)
        fe.seek(0)
        out, err = fo.read().decode("utf-8", "replace"), fe.read().decode("utf-8", "replace")
    if timed_out:
        raise TimeoutError(f"child exceeded {timeout_s:.0f} s: {err[-200:]}")
    return rc, out, err


def other_config_legs(args, sessions) -> dict:
    """The other single-GPU shapes under the driver's clock: after the headline has been measured (and
    is already in `out`), short legs in FRESH child processes, each under a time limit so that none can cost the
    headline -- the cpu_baseline pattern, never a re-exec.  Same code path as the headline (this file, other
    arguments); the whole-job leg is `--whole-job` restricted to the two modes a caller of the reference uses."""
    import subprocess

    for sp in sessions:          # the children want the HBM (config 5's trees: 2 x 4 096 slots x 8 608 blocks x 128 B = 9 GB)
        sp.close()
    torch.cuda.empty_cache()
    base = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-configs"]
    legs = {
        "config4": base + ["--blocks", "8", "--channels", "64", "--n-mcts", "800", "--games-per-gpu", "4096", "--steps", "3", "--warmup", "1"],
        "config5_per_gpu": base + ["--blocks", "8", "--channels", "64", "--n-mcts", "200", "--games-per-gpu", "8192", "--steps", "3", "--warmup", "1"],
        "config5_per_gpu_dirichlet": base + ["--blocks", "8", "--channels", "64", "--n-mcts", "200", "--games-per-gpu", "8192", "--steps", "3", "--warmup", "1",
                                             "--dirichlet", "1.0,0.25"],
        "config2_eval_cache": base + ["--steps", "6", "--warmup", "1", "--eval-cache", str(1 << 23)],
        "reference_default_job": base + ["--whole-job", "--whole-job-modes", "device_mode,numpy_callback,extension_eval_cache_device_mode"],
        "product_loop": base + ["--product-legs"],
        "config1": base + ["--config1-only"],
        # the north star's ">= 70 % of the HBM roofline on select/backup" is a question about LARGE launches: the stand-alone step
        # kernel alone on the chip at 16 384 .. 131 072 games per launch, uniform evaluator, trees grown to steady state
        "tree_kernel_sweep": [sys.executable, os.path.join(ROOT, "tools", "tree_roofline.py"), "--games", "16384,65536,131072", "--steps", "100", "--preroll", "1500"],
    }
    res = {}
    t_all = time.perf_counter()
    for name, cmd in legs.items():
        t0 = time.perf_counter()
        left = args.other_configs_total_seconds - (t0 - t_all)     # the legs TOGETHER are bounded too
        if left < 20.0:
            res[name] = {"error": f"skipped: the legs' total budget of {args.other_configs_total_seconds:.0f} s is spent", "leg_seconds": 0.0}
            continue
        try:
            rc, stdout, stderr = run_child(cmd, min(args.other_configs_seconds, left))
            line = [l for l in stdout.splitlines() if l.startswith("{")]
            if rc != 0 or not line:
                raise RuntimeError(f"child exited with {rc}: {stderr[-300:]}")
            d = json.loads(line[-1])
            if name == "tree_kernel_sweep":
                res[name] = {"kernel": d["kernel"], "evaluator": d["evaluator"], "peak_GBps": d["peak_GBps"],
                             "by_games_per_launch": {str(r["games_per_launch"]): {"device_clock_us": r["device_clock_us"], "algorithmic_bytes_per_launch": r["algorithmic_bytes_per_launch"],
                                                                                   "GBps": r["GBps_device_clock"], "frac_of_hbm_peak": r["frac_device_clock"]} for r in d["sweep"]},
                             "note": "north_star asks for >= 0.70 on select/backup: not reached at any launch size (DESIGN.md 4.2: the kernel is bound by instruction issue and dependent-load latency, not bytes)"}
            elif name in ("config1", "product_loop"):
                res[name] = d
            elif name == "reference_default_job":
                keys = ("games_per_s", "sims_per_s", "seconds", "steps", "samples", "pickle_seconds", "unpickle_seconds", "pickle_bytes", "pickle_round_trip_identical",
                        "games_per_s_play_plus_pickle", "phases_s")
                res[name] = {"workload": d["config"]["workload"],
                             "device": {k: d["device_mode"][k] for k in keys},
                             "numpy_callback": {k: d["numpy_callback"][k] for k in keys + ("samples_identical_to_device_mode",)},
                             "extension_eval_cache_device": {k: d["extension_eval_cache_device_mode"][k] for k in keys + ("samples_identical_to_device_mode", "cache_hit_rate")}}
            else:
                res[name] = {"workload": d["config"]["workload"], "games_per_s": d["value"], "sims_per_s": d["sims_per_s"], "ms_per_round": d["ms_per_round"],
                             "games_completed": d["games_completed"], "timed_rounds": d["steps"] * d["config"]["rounds_per_step"],
                             "nn_tflops": d["nn"]["achieved"], "step_kernel_us": d["roofline"]["step_alone"]["device_clock"]["avg_kernel_us"]}
                if d.get("eval_cache"):
                    res[name]["eval_cache"] = d["eval_cache"]
                    res[name]["note"] = "EXTENSION (evaluation cache, off by default): steady state like the headline; sample identity is asserted by product_loop.eval_cache_10xG on a whole job"
            res[name]["leg_seconds"] = time.perf_counter() - t0
        except Exception as e:   # a leg never costs the headline
            res[name] = {"error": repr(e)[:400], "leg_seconds": time.perf_counter() - t0}
    return res


def launch_ranks(n: int, real_stdout: int) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N rank processes here, one
    per GPU, exactly as `torch.distributed.run --nproc-per-node N` would (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*), relay rank 0's JSON line and return the worst exit code.  The children are FRESH processes
    created before this one has made any GPU call (never a re-exec of a process that initialised the GPU);
    if a rank dies the others are stopped (by pid) instead of waiting in a barrier for ever."""
    import socket
    import subprocess
    import tempfile

    if os.environ.get("C4_BENCH_SAME_DEVICE") != "1":
        have = torch.cuda.device_count()          # counts devices without initialising one
        if n > have:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} device(s) visible\n")
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = tempfile.TemporaryFile() if r == 0 else subprocess.DEVNULL
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, cwd=os.getcwd()))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        time.sleep(0.2)
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):      # a rank failed: the others would block in their next collective
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if rcs[i] is None:
                    try:
                        rcs[i] = p.wait(timeout=30)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[i] = p.wait()
    outs[0].seek(0)
    lines = [l for l in outs[0].read().decode(errors="replace").splitlines() if l.startswith("{")]
    for l in lines:
        os.write(real_stdout, (l + "\n").encode())
    worst = max(abs(rc) for rc in rcs)
    return worst if worst else (0 if lines else 1)


def main():
    # Only the JSON line may reach stdout: libraries (RCCL prints a version banner) write to the
    # process's fd 1 behind Python's back, so fd 1 is pointed at stderr for the whole run and the
    # result goes to a private duplicate of the original stdout.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    if os.environ.get("C4_BENCH_WATCHDOG"):   # debugging aid: dump every thread's Python stack and exit after N seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["C4_BENCH_WATCHDOG"]), exit=True, file=sys.stderr)

    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20,
                    help="timed steps of --rounds-per-step lock-step rounds each; 20 x 704 rounds complete >= 10 x 4096 games (SURVEY 8d, C2)")
    ap.add_argument("--warmup", type=int, default=1, help="untimed steps after the pre-roll")
    ap.add_argument("--rounds-per-step", type=int, default=704,
                    help="lock-step rounds (one MCTS simulation per resident game) per bench step; 704 = 11 replays of the 64-round HIP graph")
    ap.add_argument("--games-per-gpu", type=int, default=4096, help="resident games per GPU (BASELINE config 2: 4096)")
    ap.add_argument("--n-mcts", type=int, default=100)
    ap.add_argument("--blocks", type=int, default=4, help="residual blocks")
    ap.add_argument("--channels", type=int, default=32)
    ap.add_argument("--preroll", type=int, default=-1, help="untimed steps to reach steady state (-1 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=20.0)
    ap.add_argument("--eager", action="store_true", help="no HIP graph: launch every kernel from the host")
    ap.add_argument("--steps-per-graph", type=int, default=64, help="lock-step rounds per HIP-graph replay (paired graph, round 3: 16 -> 24.8 k, 32 -> 25.2 k, 64 -> 25.5 k games/s)")
    ap.add_argument("--one-sim-per-step", action="store_true",
                    help="A/B knob: C4_FLAG_ONE_SIM_PER_STEP (no same-launch simulation for terminal leaves)")
    ap.add_argument("--instrumented-steps", type=int, default=320, help="event-bracketed step-kernel launches for the roofline object (>= 300 whatever --steps is)")
    ap.add_argument("--eval-cache", type=int, default=0,
                    help="EXTENSION, off by default and NOT part of the headline: evaluation-cache entries per session "
                         "(c4_session_set_eval_cache); repeated positions then skip the evaluator")
    ap.add_argument("--eval-cache-sims", type=int, default=0, help="simulations per game per launch with the cache (0 = 6)")
    ap.add_argument("--dirichlet", default="", help="ALPHA,EPS: Dirichlet root noise (build extension named by BASELINE config 5; the reference has none), off by default")
    ap.add_argument("--sessions", type=int, default=2,
                    help="the resident games are split over this many sessions that replay their HIP graphs "
                         "concurrently on separate streams (1 = one session, one stream)")
    ap.add_argument("--independent-graphs", action="store_true",
                    help="A/B knob: two sessions replay two independent graphs (round 1-2) instead of ONE graph that pipelines them "
                         "explicitly (session.capture_pair)")
    ap.add_argument("--gemm", default=None, choices=["hip", "hipblaslt"],
                    help="A/B knob: hidden-layer GEMM backend (default: the hand-written batch-invariant MFMA GEMM)")
    ap.add_argument("--gemm-config", default=None, help="A/B knob: c4_linear_bf16 tile configuration, N or 'wide,narrow' (0 = automatic)")
    ap.add_argument("--no-fused-step", action="store_true", help="A/B knob: output kernel and step kernel as two launches (round 3) instead of one")
    ap.add_argument("--pair-offset", type=int, default=1, help="A/B knob: capture_pair's offset_stage (session B starts when this stage of A's first round is done)")
    ap.add_argument("--tower-config", type=int, default=0, help="A/B knob: c4_conv_tower_bf16 workgroup shape (0 = automatic)")
    ap.add_argument("--stream-priorities", default="", help="A/B knob: HIP stream priorities of the sessions' streams, e.g. '-1,0' (lower = higher priority)")
    ap.add_argument("--no-loader-waves", action="store_true", help="A/B knob: round 3's small-batch GEMM tiles (27 / 9 / 23 / 10) instead of their wave-specialised forms (41 / 42 / 44 / 43) up to 1 024 rows")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="internal: compute the cpu_baseline object alone and print it (the bench runs this leg in a child "
                         "process under a time limit, so that the checker can never cost the GPU line)")
    ap.add_argument("--product-legs", action="store_true", help="internal: whole play_games calls at BASELINE config 2's shape (10 x G games; one generation; with the evaluation cache), own JSON line")
    ap.add_argument("--config1-only", action="store_true", help="internal: BASELINE config 1 (32 games, n = 10, 1-block net) on the host cores and on the GPU, own JSON line")
    ap.add_argument("--whole-job", action="store_true",
                    help="instead of the steady-state bench: the reference's default self-play job, whole, in callback and device modes (own JSON line)")
    ap.add_argument("--whole-job-modes", default="", help="comma-separated subset of the --whole-job modes (default: all; device_mode first: the others are compared with it)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short legs for the other single-GPU shapes (BASELINE config 4, config 5's per-GPU share, the reference's default job) that the default N = 1 run attaches as `other_configs`")
    ap.add_argument("--other-configs-seconds", type=float, default=150.0, help="time limit of EACH other_configs leg (a child process)")
    ap.add_argument("--other-configs-total-seconds", type=float, default=480.0, help="time limit of all other_configs legs TOGETHER (legs past it are skipped)")
    ap.add_argument("--whole-job-games", type=int, default=1700)
    ap.add_argument("--whole-job-n-mcts", type=int, default=1400)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: this process becomes one (it has not touched the GPU yet, and never will)
        sys.exit(launch_ranks(args.gpus, real_stdout))
    if args.gpus != world:
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device")
    # C4_BENCH_SAME_DEVICE=1 (test knob, with C4_BENCH_BACKEND=gloo): every rank on device 0, so the whole
    # N > 1 code path can be exercised on a one-GPU box
    dev_index = 0 if os.environ.get("C4_BENCH_SAME_DEVICE") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if args.whole_job:
        return whole_job(args, device, real_stdout)
    if args.product_legs:
        return product_legs(args, device, real_stdout)
    if args.config1_only:
        os.write(real_stdout, (json.dumps(config1_leg(device)) + "\n").encode())
        return
    if args.cpu_baseline_only:
        from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig
        torch.manual_seed(1337)
        net = InferenceNet(ConnectFourNet(ModelConfig(args.blocks, args.channels, 4, 2)), device, dtype=torch.bfloat16)
        out = cpu_baseline(net, device, args.n_mcts, min(usable_cores(), 64), args.cpu_baseline_seconds)
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
        return
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig, flops_per_leaf
    from c4a0_amd.session import DeviceSession

    G, n_iter = args.games_per_gpu, args.n_mcts
    if args.no_fused_step:
        DeviceSession.fuse_output_step = False
    cfg = ModelConfig(args.blocks, args.channels, 4, 2)
    torch.manual_seed(1337)
    net = InferenceNet(ConnectFourNet(cfg), device, dtype=torch.bfloat16, gemm=args.gemm, gemm_config=args.gemm_config,
                       tower_config=args.tower_config)
    if args.no_loader_waves:
        net.use_loader_waves = False
    if max(1, min(args.sessions, args.games_per_gpu)) == 1 and not args.eager:
        net.latency_mode = True      # as api._play: one session alone on the device picks the tiles measured for that (nn.InferenceNet.latency_mode)

    P = 1 if args.eager else max(1, min(args.sessions, G))
    R = max(1, args.rounds_per_step)
    timed_rounds, warm_rounds = args.steps * R, args.warmup * R
    preroll = args.preroll if args.preroll >= 0 else int(2.0 * 15.0 * n_iter)  # ~2 game lengths of sims
    total_steps = preroll + warm_rounds + timed_rounds + 2 * max(300, args.instrumented_steps) + 64
    sims_per_game_lo = 8 * n_iter  # generous lower bound on sims per game -> upper bound on games needed
    trips = (args.eval_cache_sims or 6) if args.eval_cache else 1   # the cache lets a game run several simulations per step
    n_games = int(G * (2 + trips * total_steps / sims_per_game_lo)) + G
    # ids sharded id % world == rank: rank r plays ids r, r+W, ...; on the GPU the resident
    # games are split over P sessions (c4a0_amd.session.run_sessions explains why), session p taking
    # every P-th of the rank's requests
    ids = [rank + world * i for i in range(n_games)]
    sessions, streams, graphs, graphs1 = [], [], [], []   # graphs1: one round per replay, for the K % U remainder
    U = 1 if args.eager else max(1, args.steps_per_graph)
    paired = P == 2 and not args.eager and not args.independent_graphs
    for p in range(P):
        sp = DeviceSession((G + P - 1 - p) // P, n_iter, 6.6, 0.01, device=device, planes_dtype=torch.bfloat16,
                           one_sim_per_step=args.one_sim_per_step)
        sp.set_games([(i, 0, 0) for i in ids[p::P]])
        if args.eval_cache:
            sp.set_eval_cache(args.eval_cache, args.eval_cache_sims)
        if args.dirichlet:
            sp.set_dirichlet(*[float(v) for v in args.dirichlet.split(",")])
        prio = [int(v) for v in args.stream_priorities.split(",")] if args.stream_priorities else []   # A/B knob
        st = (torch.cuda.Stream(device=device, priority=prio[p % len(prio)]) if prio else torch.cuda.Stream(device=device)) if P > 1 else torch.cuda.current_stream(device)
        with torch.cuda.stream(st):
            sp.bind(st)
            sp.start()
        st.synchronize()
        if args.eager:
            sp.set_timing(False)
            graphs.append(None)
            graphs1.append(None)
        elif not paired:
            graphs.append(sp.capture_steps(net, U, stream=st if P > 1 else None))
            graphs1.append(sp.capture_steps(net, 1, stream=st if P > 1 else None) if U > 1 else graphs[-1])
        sessions.append(sp)
        streams.append(st)
    pair_graph = pair_graph1 = None
    if paired:   # both sessions' rounds in ONE graph, explicitly pipelined against each other (session.capture_pair)
        from c4a0_amd.session import capture_pair
        pair_graph = capture_pair(sessions, streams, net, U, offset_stage=args.pair_offset)
        pair_graph1 = capture_pair(sessions, streams, net, 1, offset_stage=args.pair_offset) if U > 1 else pair_graph

    # RCCL comes up only now, AFTER the HIP graphs are captured: its watchdog thread must not poll
    # events while a stream capture is open
    dist = None
    if world > 1 or os.environ.get("C4_BENCH_FORCE_DIST") == "1":  # the env knob exercises the RCCL path at world size 1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")   # only reached without a launcher (the world-size-1 knob above)
        backend = os.environ.get("C4_BENCH_BACKEND", "nccl")   # "nccl" = RCCL over xGMI; "gloo" only for the one-GPU test
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    red_dev = device if (dist is None or dist.get_backend() == "nccl") else torch.device("cpu")   # where small reductions live

    def counters():
        tot = {}
        for sp in sessions:
            for k, v in sp.counters().items():   # synchronises that session's stream
                tot[k] = max(tot.get(k, 0), v) if k in ("error", "error_slot") else tot.get(k, 0) + v
        return tot

    def run_steps(k):
        """k lock-step rounds of every session: HIP-graph replays of U rounds each, remainder launched eagerly."""
        if paired:
            with torch.cuda.stream(streams[0]):
                for _ in range(k // U):
                    pair_graph.replay()
                for _ in range(k % U):
                    pair_graph1.replay()
            return
        for _ in range(k // U if graphs[0] is not None else 0):
            for st, g in zip(streams, graphs):
                with torch.cuda.stream(st):
                    g.replay()
        for _ in range(k % U if graphs[0] is not None else k):
            for sp, st, g1 in zip(sessions, streams, graphs1):
                with torch.cuda.stream(st):
                    if g1 is not None:
                        g1.replay()
                    else:
                        sp.evaluate(net)
                        sp.step()

    run_steps(preroll)
    run_steps(warm_rounds)
    c0 = counters()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(timed_rounds)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    c1 = counters()
    if c1["error"]:
        sys.exit(f"device error {c1['error']} in slot {c1['error_slot']}")
    d = {k: c1[k] - c0[k] for k in c1 if k not in ("error", "error_slot")}
    elapsed = t1 - t0

    # ---- instrumented segment right after the timed steps (same steady state): the step kernel of
    # session 0 bracketed by HIP events on its stream, and timed on the device clock inside the
    # kernel; the other sessions wait.  (Inside the graph-replayed region nothing can be bracketed
    # per launch.)
    n_inst = max(300, args.instrumented_steps)
    sess, st0 = sessions[0], streams[0]
    sess.set_timing(True)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_inst)]
    ci0 = sess.counters()
    with torch.cuda.stream(st0):
        for a_ev, b_ev in ev:
            sess.evaluate(net)
            a_ev.record(st0)
            sess.step()
            b_ev.record(st0)
    torch.cuda.synchronize()
    ci1 = sess.counters()
    if ci1["error"]:
        sys.exit(f"device error {ci1['error']} in slot {ci1['error_slot']}")
    di = {k: ci1[k] - ci0[k] for k in ci1 if k not in ("error", "error_slot")}
    step_kernel_ms = sum(a_ev.elapsed_time(b_ev) for a_ev, b_ev in ev)
    # ---- ... and the round AS THE TIMED REGION LAUNCHES IT (per-launch timing off: the heads' output layers and the step are one
    # launch, c4_out_step_kernel), every launch of session 0 bracketed by HIP events on its stream
    timeline, tl_rounds, dt_tl, null_us = None, 0, None, None
    if net.path == "hip" and not args.eager and not args.no_fused_step and not args.eval_cache and not args.dirichlet and rank == 0:
        ct0 = sess.counters()
        timeline, tl_rounds, null_us = kernel_timeline(sess, st0, net, n_rounds=max(300, args.instrumented_steps))
        ct1 = sess.counters()
        if ct1["error"]:
            sys.exit(f"device error {ct1['error']} in slot {ct1['error_slot']}")
        dt_tl = {k: ct1[k] - ct0[k] for k in ct1 if k not in ("error", "error_slot")}
    if any(sp.counters()["games_started"] >= sp.n_games for sp in sessions):
        sys.exit("bench ran out of queued games; raise n_games")

    # ---- multi-GPU: the one exchange step of the path -- all-gather the finished samples (untimed
    # end-of-job step; its time is reported beside the throughput).  Exactly what
    # c4a0_amd.distributed.play_games_sharded runs: pack on the device, counts + padded records
    # all-gathered, shards merged into request order -- and the merged result is checked.
    allgather = None
    if dist is not None:
        try:
            from c4a0_amd.api import merge_parts
            from c4a0_amd.distributed import gather_shards, merge_shards
            torch.cuda.synchronize()
            dist.barrier()
            tg0 = time.perf_counter()
            pieces = [(np.arange(p, n_games, P, dtype=np.int64), sp.sample_counts(), sp.pack_samples_device()) for p, sp in enumerate(sessions)]
            local, local_counts = merge_parts(n_games, pieces) if P > 1 else (pieces[0][2], pieces[0][1])
            per_rank, per_counts = gather_shards(local, local_counts, n_games * world)
            merged, all_counts = merge_shards(per_rank, per_counts, n_games * world)
            torch.cuda.synchronize()
            tg1 = time.perf_counter()
            # request position g = i * world + rank carries game id rank + world * i = g
            cnt = torch.as_tensor(all_counts.astype(np.int64), device=merged.device)
            want_ids = torch.repeat_interleave(torch.arange(n_games * world, device=merged.device), cnt)
            got_ids = merged[:, :8].contiguous().view(torch.int64).reshape(-1)
            n_fin = int((cnt > 0).sum().item())
            done_all = torch.tensor([float(sum(sp.counters()["games_done"] for sp in sessions))], dtype=torch.float64, device=red_dev)
            dist.all_reduce(done_all, op=dist.ReduceOp.SUM)
            ok = bool(torch.equal(got_ids, want_ids)) and merged.shape[0] == int(cnt.sum().item()) and n_fin == int(done_all.item())
            allgather = {"ms": (tg1 - tg0) * 1e3, "records_per_rank": [int(p_.shape[0]) for p_ in per_rank],
                         "bytes_total": int(sum(p_.numel() for p_ in per_rank)), "games_merged": n_fin,
                         "merged_in_request_order_and_complete": ok}
        except Exception as e:  # never lose the throughput line to the epilogue
            allgather = {"error": repr(e)}

    games, sims, elapsed_max = float(d["games_done"]), float(d["sims"]), elapsed
    per_rank = None
    if dist is not None:
        mine = torch.tensor([games, elapsed], dtype=torch.float64, device=red_dev)   # per-rank view, for reading the scaling curve
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"games_completed": [float(x[0]) for x in allr], "elapsed_s": [float(x[1]) for x in allr]}
        t = torch.tensor([games, sims, float(d["ref_skipped_sims"])], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        m = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        games, sims, skipped = t.tolist()
        elapsed_max = float(m.item())
    else:
        skipped = float(d["ref_skipped_sims"])

    if rank == 0:
        ab = algorithmic_bytes(di, 2)
        avg_kernel_s = step_kernel_ms / 1e3 / n_inst                       # HIP events around each launch
        achieved = ab["total"] / n_inst / avg_kernel_s / 1e9
        # the same launches on the device clock (first wavefront start -> last wavefront end,
        # s_memrealtime stamps taken inside the kernel): what rocprofv3's kernel duration measures
        dev_s = di["step_kernel_ns"] / 1e9 / max(1, di["step_launches"])
        achieved_dev = ab["total"] / n_inst / max(dev_s, 1e-12) / 1e9
        traffic, traffic_current, traffic_fused = None, None, None
        tpath = os.path.join(ROOT, "profiles", "step_kernel_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_fused = (tj.get("c4_out_step_kernel") or {}).get("hbm_bytes_per_launch")
                # counters are collected in their own rocprofv3 --pmc passes (tools/profile/run_r06.sh), not in this run: say whether
                # they were taken on the step kernel this library holds (hash of c4_session.hip + c4_device.hpp at collection time)
                traffic_current = tj.get("step_kernel_source_hash") == step_kernel_source_hash()
            except Exception:
                traffic = None
        fl = flops_per_leaf(cfg)
        mfma_busy, mfma_src = None, None   # counters under the evaluator: not re-measured here, read from the committed PMC summary
        try:
            pmc_file = next(f for f in ("r06_evaluator_pmc.json", "r05_evaluator_pmc.json", "r04_evaluator_pmc.json", "r03_evaluator_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pm = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            mfma_busy = pm["backends"]["hip0" if net.gemm == "hip" else "hipblaslt0"]["evaluator_mfma_busy_frac_of_chip_time_weighted"]
            mfma_src = (f"profiles/{pmc_file}: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles), time-weighted over the tower, the three "
                        "hidden-layer GEMMs and the output kernel, each launch ALONE on the chip at 2 048 rows (rocprofv3 --pmc serialises kernels); "
                        "not re-measured in this run")
        except Exception:
            pass
        # ---- the fused launch of the timed region and the evaluator's kernels, from the event-bracketed rounds above
        rows = sess.rows
        F = 42 * cfg.conv_filter_size
        conv_flops = 2 * 42 * 9 * 2 * cfg.conv_filter_size + cfg.n_residual_blocks * 2 * (2 * 42 * 9 * cfg.conv_filter_size ** 2)
        kernels, fused = {}, None
        if timeline:
            flops = {"tower": rows * conv_flops, "gemm_first_hidden": 2 * rows * F * 2 * F, "gemm_narrow": 2 * rows * F * F}
            names = {"tower": "c4_conv_tower_kernel", "gemm_first_hidden": "c4_head_gemm_kernel (the merged 2F-wide first hidden layer of both heads)",
                     "gemm_narrow": "c4_head_gemm_kernel (an F-wide hidden layer)", "out_step": "c4_out_step_kernel"}
            ab_tl = algorithmic_bytes(dt_tl, 2)
            for label, ms, launches in timeline:
                us = ms * 1e3 / launches
                k = {"kernel": names[label], "calls_per_round": launches / tl_rounds, "avg_us": us, "launches_measured": launches}
                if label in flops:
                    tf = flops[label] / (us * 1e-6) / 1e12
                    k.update({"bound": "mfma", "flops_per_launch": flops[label], "achieved_tflops": tf, "frac_of_mfma_peak": tf / MFMA_BF16_PEAK_TFLOPS})
                else:
                    gbps = ab_tl["total"] / tl_rounds / (us * 1e-6) / 1e9
                    k.update({"bound": "hbm", "algorithmic_bytes_per_launch": ab_tl["total"] / tl_rounds, "achieved_gbps": gbps, "frac_of_hbm_peak": gbps / HBM_PEAK_GBPS})
                    fused = k
                kernels[label] = k
            # the committed rocprofv3 --kernel-trace --stats summary of THIS command (tools/profile/run_r06.sh): per-kernel durations without the
            # dispatch gap; one c4_head_gemm_kernel instance serves both layer widths at this size, so its average pools three launches
            rocprof = {}
            try:
                import csv
                stats_file = next(f for f in ("r06_kernel_stats.csv", "r05_kernel_stats.csv") if os.path.exists(os.path.join(ROOT, "profiles", f)))
                for row in csv.DictReader(open(os.path.join(ROOT, "profiles", stats_file))):
                    for key, sub in (("tower", "c4_conv_tower_kernel"), ("out_step", "c4_out_step_kernel"), ("gemm_pooled", "c4_head_gemm_kernel")):
                        if sub in row["Name"] and key not in rocprof:
                            rocprof[key] = float(row["AverageNs"]) / 1e3
                for key in ("tower", "out_step"):
                    if key in kernels and key in rocprof:
                        kernels[key]["rocprof_avg_us"] = rocprof[key]
                if "gemm_pooled" in rocprof and "gemm_first_hidden" in kernels and "gemm_narrow" in kernels:
                    g1, g2 = kernels["gemm_first_hidden"], kernels["gemm_narrow"]
                    n_l = g1["launches_measured"] + g2["launches_measured"]
                    fl_pooled = (g1["flops_per_launch"] * g1["launches_measured"] + g2["flops_per_launch"] * g2["launches_measured"]) / n_l
                    us = (g1["avg_us"] * g1["launches_measured"] + g2["avg_us"] * g2["launches_measured"]) / n_l
                    kernels["gemm_pooled"] = {"kernel": "c4_head_gemm_kernel, all three hidden-layer launches of a round pooled (as rocprofv3 names them: one instance)",
                                              "calls_per_round": n_l / tl_rounds, "avg_us": us, "rocprof_avg_us": rocprof["gemm_pooled"], "flops_per_launch": fl_pooled,
                                              "frac_of_mfma_peak": fl_pooled / (us * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                              "rocprof_frac_of_mfma_peak": fl_pooled / (rocprof["gemm_pooled"] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS}
                if "tower" in kernels and "rocprof_avg_us" in kernels["tower"]:
                    kernels["tower"]["rocprof_frac_of_mfma_peak"] = kernels["tower"]["flops_per_launch"] / (kernels["tower"]["rocprof_avg_us"] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS
                if fused and "rocprof_avg_us" in fused:
                    fused["rocprof_frac_of_hbm_peak"] = fused["algorithmic_bytes_per_launch"] / (fused["rocprof_avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
                kernels["rocprof_source"] = f"profiles/{stats_file} (rocprofv3 --kernel-trace --stats of this command; not re-measured in this run)"
            except Exception:
                pass
            kernels["one_element_kernel_interval_us"] = null_us
            per_round_us = sum(ms for _l, ms, _n in timeline) * 1e3 / tl_rounds
            kernels["sum_per_session_round_us"] = per_round_us
            kernels["share_of_round"] = {label: ms * 1e3 / tl_rounds / per_round_us for label, ms, _n in timeline}
            kernels["rows_per_launch"] = rows
            kernels["method"] = (f"{tl_rounds} eager rounds of one session ({rows} rows) right after the timed region, a HIP event on its stream at every launch boundary, queued in batches "
                                 "behind a blocker so that the GPU runs them back to back; the other session idle (rocprofv3's kernel trace serialises the two queues the same way: "
                                 "profiles/r06_kernel_stats.csv holds its averages for the same kernels: rocprof_avg_us); an event-to-event interval is the kernel PLUS the dispatch of a dependent launch -- the same "
                                 "bracket around a chain of one-element kernels gives one_element_kernel_interval_us -- which is why avg_us exceeds rocprof_avg_us by 2.5-3.5 us")
        issue = None   # the step kernel's issue-slot occupancy, from the committed counter summary (own rocprofv3 --pmc passes, tools/profile/run_r06.sh)
        try:
            pmc_step = next(f for f in ("r06_step_kernel_pmc.json", "r05_step_kernel_pmc.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pw = json.load(open(os.path.join(ROOT, "profiles", pmc_step)))["games_65536"]["per_wave"]
            waves_per_simd = 4   # 127-128 VGPRs: four wavefronts share a SIMD
            issue = {"issue_slots_taken": pw["SQ_ACTIVE_INST_ANY_per_wave"] * waves_per_simd / pw["SQ_WAVE_CYCLES_per_wave"],
                     "valu_share_of_issue": pw["SQ_ACTIVE_INST_VALU_per_wave"] / pw["SQ_ACTIVE_INST_ANY_per_wave"],
                     "valu_pipe_busy": pw["SQ_ACTIVE_INST_VALU_per_wave"] * waves_per_simd / pw["SQ_WAVE_CYCLES_per_wave"],
                     "insts_per_wavefront": {k: pw[f"SQ_INSTS_{k}_per_wave"] for k in ("VALU", "SALU", "LDS", "VMEM_RD", "VMEM_WR")},
                     "waves_per_simd": waves_per_simd, "at_games_per_launch": 65536,
                     "formula": "issue_slots_taken = SQ_ACTIVE_INST_ANY x wavefronts per SIMD / SQ_WAVE_CYCLES (both in quad-cycles per wavefront): the share of a SIMD's cycles in which one of its "
                                "wavefronts issues (an upper bound on the issue stage's load: different instruction types of different wavefronts can issue together); valu_pipe_busy = the same "
                                "with SQ_ACTIVE_INST_VALU: the vector ALU's own occupancy",
                     "source": f"profiles/{pmc_step} (rocprofv3 --pmc, own passes; stand-alone c4_step_kernel = the step_body the fused launch runs); not re-measured in this run"}
        except Exception:
            pass
        shape = (G, n_iter, cfg.n_residual_blocks, cfg.conv_filter_size)
        named = {(4096, 100, 4, 32): "BASELINE config 2 per GPU", (4096, 800, 8, 64): "BASELINE config 4",
                 (8192, 200, 8, 64): "BASELINE config 5's per-GPU share (temperature schedule on, Dirichlet noise " + (f"alpha,eps = {args.dirichlet}" if args.dirichlet else "off") + ")"
                 }.get(shape, "custom shape (not a BASELINE configuration)")
        out = {
            "metric": f"self-play games/sec (and MCTS sims/sec) at n_mcts={n_iter}",
            "value": games / elapsed_max,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "ms_per_round": elapsed_max / timed_rounds * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{named}: {G} concurrent games, n_mcts_iterations={n_iter}, "
                                   f"{cfg.n_residual_blocks}-block/{cfg.conv_filter_size}-ch ResNet bf16, c_exploration=6.6, c_ply_penalty=0.01",
                       "rounds_per_step": R, "step": f"{R} lock-step rounds (one MCTS simulation per resident game each)",
                       "games_per_gpu": G, "n_mcts_iterations": n_iter, "parallelism": f"games sharded id%{world}",
                       "evaluator": "eager" if args.eager else (f"one hip-graph x{U} rounds of both sessions, explicitly pipelined (session.capture_pair)" if paired
                                                                else f"hip-graph x{U} steps (evaluator + step kernel)"),
                       "concurrent_sessions": P, "games_per_session": [sp.n_slots for sp in sessions], "preroll_steps": preroll,
                       "eval_cache_entries_per_session": args.eval_cache, "dirichlet_alpha_eps": args.dirichlet or None,
                       "tree_dtype": "u64 bitboards + f32 UCT"},
            "sims_per_s": sims / elapsed_max,
            "ref_equivalent_sims_per_s": (sims + skipped) / elapsed_max,
            "games_completed": games,
            "eval_cache": ({"probes": d["eval_cache_probes"], "hits": d["eval_cache_hits"],
                            "hit_rate": d["eval_cache_hits"] / max(1, d["eval_cache_probes"]),
                            "note": "EXTENSION switched on by --eval-cache: repeated positions skip the evaluator; not the headline configuration"}
                           if args.eval_cache else None),
            "sims_per_game": sims / max(1.0, games),
            "roofline": {"bound": "hbm",
                         "kernel": ("c4_out_step_kernel: the launch of the timed region -- the heads' output layers, then expand + backup + move + select + encode of the same games"
                                    if fused else "c4_step_kernel (expand+backup+move+select+encode, fused)"),
                         "measured_on": ("the event-bracketed rounds right after the timed region (see kernels.method); `achieved` = SURVEY 8(d)'s tree bytes (92 S + 28 K + 332 E + encode, "
                                         "device-counted) per launch / this launch's average duration -- the output layers' own operands (head_out_operand_bytes_per_launch) are not counted"
                                         if fused else "the instrumented stand-alone launches after the timed region"),
                         "achieved": fused["achieved_gbps"] if fused else achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (fused["achieved_gbps"] if fused else achieved) / HBM_PEAK_GBPS,
                         "avg_launch_us": fused["avg_us"] if fused else avg_kernel_s * 1e6,
                         "launches_measured": fused["launches_measured"] if fused else n_inst, "games_per_launch": sess.rows,
                         "algorithmic_bytes_per_launch": fused["algorithmic_bytes_per_launch"] if fused else ab["total"] / n_inst,
                         "head_out_operand_bytes_per_launch": (rows * 2 * F * 2 + 9 * F * 2 + rows * 36) if fused else None,
                         "traffic": traffic_fused if fused else traffic,
                         "traffic_source": "profiles/step_kernel_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, each in its own run, corrected per MI355X_MICROARCH.md); not re-measured in this run",
                         "traffic_collected_on_this_step_kernel": traffic_current,
                         # what binds it: not bytes.  At every launch size the SIMDs' issue slots are what is full (VALU above all), see DESIGN.md 4.2
                         "binding_roof": "valu_issue", "issue_slot_occupancy": issue,
                         "step_alone": {
                             "kernel": "c4_step_kernel (the same step_body as a launch of its own: per-launch device-clock timing needs it; what the callback mode and eager sessions run)",
                             "achieved": achieved, "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                             # the memory system's ceiling for THIS access pattern (random whole 128-byte lines from an HBM-resident table,
                             # tools/gather_lab.hip): 6.7 TB/s, i.e. 84 % of the 8 TB/s the fraction above is quoted against
                             "frac_of_practical_ceiling": achieved / 6700.0,
                             "device_clock_frac_of_practical_ceiling": achieved_dev / 6700.0,
                             "avg_launch_us": avg_kernel_s * 1e6,
                             "device_clock": {"avg_kernel_us": dev_s * 1e6, "achieved": achieved_dev, "frac": achieved_dev / HBM_PEAK_GBPS,
                                              "note": "in-kernel s_memrealtime stamps; the HIP-event bracket adds the dispatch and completion latency of one launch"},
                             "launches_measured": n_inst, "algorithmic_bytes_per_launch": ab["total"] / n_inst},
                         "practical_ceilings": {"random_128B_lines_from_HBM_GBps": 6700, "dependent_line_chain_ns_per_level_idle": 550,
                                                "source": "profiles/r02_gather_lab.jsonl (tools/gather_lab.hip on MI355X): what the memory system gives the tree walk's access pattern; at this launch size the kernel is bound by one wavefront's dependent chain, not by bytes (DESIGN.md 4.2)"},
                         "bytes_per_sim": {k: v / max(1, di["sims"]) for k, v in ab.items()},
                         "S_per_sim": di["select_levels"] / max(1, di["sims"]), "K_per_sim": di["backup_nodes"] / max(1, di["sims"]),
                         "E_per_sim": di["expansions"] / max(1, di["sims"]),
                         "kernels": kernels or None},
            "nn": {"bound": "mfma", "flops_per_leaf": fl, "achieved": fl * G * timed_rounds / elapsed / 1e12,
                   "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                   "frac": fl * G * timed_rounds / elapsed / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                   "note": "evaluator FLOPs over the WHOLE wall time of the timed steps (tree kernels and launch gaps included): a lower bound on the evaluator's own rate",
                   "mfma_busy_frac": mfma_busy, "mfma_busy_source": mfma_src, "hidden_layer_gemm": net.gemm,
                   "path": net.path},   # "hip" = hand-written tower + GEMM + output kernels; "torch" = PyTorch / library kernels somewhere in the chain
        }
        if allgather is not None:
            out["sample_allgather"] = allgather
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            # the checker's leg runs in a child process under a time limit: whatever happens to it, the GPU line is printed
            import subprocess
            for sp in sessions:
                sp.close()
            sessions = []
            torch.cuda.empty_cache()
            try:
                limit = max(120.0, 8.0 * args.cpu_baseline_seconds)
                rc, stdout, stderr = run_child([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--n-mcts", str(n_iter), "--blocks", str(args.blocks),
                                                "--channels", str(args.channels), "--cpu-baseline-seconds", str(args.cpu_baseline_seconds)], limit)
                line = [l for l in stdout.splitlines() if l.startswith("{")]
                if rc != 0 or not line:
                    raise RuntimeError(f"child exited with {rc}: {stderr[-300:]}")
                out["cpu_baseline"] = json.loads(line[-1])
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "unit": "games/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
        if world == 1 and not args.no_other_configs and shape == (4096, 100, 4, 32):
            out["other_configs"] = other_config_legs(args, sessions)
            sessions = []
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    for sp in sessions:
        sp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
