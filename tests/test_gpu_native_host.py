"""`c4_play_games_bf16` (include/c4a0_hip.h, c4a0_amd/csrc/c4_selfplay_host.hip): the body of the reference's `self_play()`
(rust/src/self_play.rs:39-129) as ONE native call -- sessions, streams, the paired HIP graph, completion polling, tail narrowing,
the merged hand-over.  It must return, byte for byte, what the Python loop of `play_games(evaluator=InferenceNet)` returns (which
the rest of the suite holds to the oracle), for every way the job can be shaped."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _net(blocks, channels, pol=4, val=2, seed=1337):
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(seed)
    return InferenceNet(ConnectFourNet(ModelConfig(blocks, channels, pol, val)), torch.device("cuda:0"), dtype=torch.bfloat16)


def _both(net, n_games, n_iter, native_kw=None, **kw):
    import c4a0_amd
    from c4a0_amd.native import play_games_native

    reqs = [c4a0_amd.GameMetadata(7000 + 3 * i, 0, 0) for i in range(n_games)]
    st_n, st_p = {}, {}
    got = play_games_native(reqs, 4096, n_iter, 6.6, 0.01, net, stats=st_n, **dict(kw, **(native_kw or {})))
    want = c4a0_amd.play_games(reqs, 4096, n_iter, 6.6, 0.01, evaluator=net, stats=st_p, host_loop="python", **kw)
    assert st_n["host_loop"] == "native" and st_p["host_loop"] == "python"
    (r1, c1), (r2, c2) = got.to_records(), want.to_records()
    assert np.array_equal(c1, c2) and r1.tobytes() == r2.tobytes()
    for k in ("sims", "select_levels", "backup_nodes", "expansions", "moves", "games_done", "samples"):
        assert st_n[k] == st_p[k], k
    assert st_n["games_done"] == n_games and st_n["error"] == 0
    return st_n, st_p


@pytest.mark.parametrize("blocks,channels,pol,val", [(1, 32, 4, 2), (2, 64, 2, 3), (1, 32, 2, 2)])
def test_one_session_small_jobs(blocks, channels, pol, val):
    """One session alone (fewer than 2 048 resident games): refill from the queue, tiles chosen for a session alone, narrowing."""
    net = _net(blocks, channels, pol, val)
    st, _ = _both(net, 200, 20, resident_games=64)
    assert st["concurrent_sessions"] == 1 and st["n_slots"] == 64
    st, _ = _both(net, 700, 12, resident_games=600)
    assert st["rows_at_end"] < 600                                  # narrowed in the tail


def test_two_paired_sessions_with_refill_and_tail():
    """The product shape in small: 4 096 resident games as two paired sessions in one graph, 6 000 games (slots refilled), the tail
    narrowed several times; also with other graph lengths (the records do not depend on them)."""
    net = _net(4, 32)
    st, st_p = _both(net, 6000, 16, resident_games=4096)
    assert st["concurrent_sessions"] == 2 and st["n_slots"] == 4096 and st["rows_at_end"] < 4096
    assert st["phases"]["graph_captures"] >= 2
    _both(net, 3000, 10, resident_games=2048, native_kw=dict(steps_per_graph=5, tail_steps_per_graph=3))


@pytest.mark.parametrize("ext", ["dirichlet", "cache", "reclaim"])
def test_extensions_and_reclaimed_arenas(ext):
    net = _net(1, 32, 2, 2)
    if ext == "dirichlet":
        _both(net, 900, 16, resident_games=800, dirichlet=(0.3, 0.25), concurrent_sessions=2)
    elif ext == "cache":
        st, _ = _both(net, 900, 16, resident_games=800, eval_cache_entries=1 << 16, concurrent_sessions=2)
        assert st["eval_cache_hits"] > 0
    else:
        n, period = 24, 4
        half = n + 2 + 8 + 2 * (2 * period * 2 + 16) + 6
        st, _ = _both(net, 700, n, resident_games=600, concurrent_sessions=2, reclaim=True, reclaim_period=period, blocks_per_slot=2 * half)
        assert st["reclaim_passes"] > 700


def test_play_games_takes_the_native_loop_by_itself_and_only_where_it_applies():
    """`play_games(evaluator=InferenceNet)` runs inside the library's loop; a subclass that overrides forward (it must see every
    evaluation), a measurement switch on the net, three sessions or a plain device callable keep the Python loop; host_loop="native"
    insists and says why it cannot."""
    import c4a0_amd
    from c4a0_amd.nn import InferenceNet
    from tests.helpers import hash_eval_torch

    net = _net(1, 32)
    reqs = [c4a0_amd.GameMetadata(i, 0, 0) for i in range(40)]
    st = {}
    base = c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=net, stats=st)
    assert st["host_loop"] == "native" and st["games_done"] == 40

    class Counting(InferenceNet):
        calls = 0

        def forward(self, *a, **k):
            Counting.calls += 1
            return super().forward(*a, **k)
        __call__ = forward

    cnet = Counting.__new__(Counting)
    cnet.__dict__.update(net.__dict__)
    for kw, ev in ((dict(), cnet), (dict(concurrent_sessions=3, resident_games=30), net), (dict(), hash_eval_torch)):
        st = {}
        res = c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=ev, stats=st, **kw)
        assert st["host_loop"] == "python"
        if ev is not hash_eval_torch:
            assert res.to_records()[0].tobytes() == base.to_records()[0].tobytes()
    assert Counting.calls > 0
    with pytest.raises(TypeError, match="host_loop='native'"):
        c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=hash_eval_torch, host_loop="native")
    with pytest.raises(ValueError):
        c4a0_amd.play_games(reqs, 64, 8, 6.6, 0.01, evaluator=net, host_loop="rust")


def test_default_shapes_and_errors():
    """No options at all: the library sizes the job itself (one session for a small job); empty jobs; multi-model requests and
    networks the kernels do not take are refused; a too-small record buffer reports what is needed."""
    import ctypes as C

    import c4a0_amd
    from c4a0_amd import _lib
    from c4a0_amd.native import network_struct, play_games_native

    net = _net(1, 32)
    st, _ = _both(net, 300, 10)
    assert st["n_slots"] == 300 and st["concurrent_sessions"] == 1
    st, st_p = _both(net, 2500, 6)                                  # one generation (no refill): ONE session although 2 048 rows would pair
    assert st["n_slots"] == 2500 and st["concurrent_sessions"] == 1 == st_p["concurrent_sessions"]
    assert play_games_native([], 64, 10, 6.6, 0.01, net).results == []
    with pytest.raises(TypeError):
        play_games_native([c4a0_amd.GameMetadata(0, 1, 2)], 64, 10, 6.6, 0.01, net)
    L, ns = _lib.lib(), network_struct(net)
    reqs = np.zeros((40, 3), dtype=np.uint64)
    reqs[:, 0] = np.arange(40)
    counts, recs, n = np.zeros(40, np.uint32), np.zeros(40, dtype=np.dtype("V64")), C.c_uint64()
    rc = L.c4_play_games_bf16(reqs.ctypes.data, 40, 8, 6.6, 0.01, C.byref(ns), None, counts.ctypes.data, recs.ctypes.data, 40, C.byref(n), None, None)
    assert rc == _lib.ERR_BAD_ARG and n.value == counts.sum() > 40 and b"room for 40" in L.c4_last_error_string()
    ns.channels = 48
    assert L.c4_play_games_bf16(reqs.ctypes.data, 40, 8, 6.6, 0.01, C.byref(ns), None, counts.ctypes.data, recs.ctypes.data, 40, C.byref(n), None, None) == _lib.ERR_BAD_ARG


def test_repeated_calls_give_back_what_they_took():
    """A training loop calls play_games once per generation (reference src/c4a0/training.py:176-189), for days: every call builds its
    sessions, streams, events and graphs and must give all of it back -- free device memory after forty jobs of either shape (one
    session; a pair with refill, narrowing and re-captures) stays where it was after the first ones, and the bytes stay the same."""
    import c4a0_amd
    from c4a0_amd.native import play_games_native

    net = _net(1, 32)
    shapes = [(300, 10, {}), (1500, 6, {"resident_games": 1024, "concurrent_sessions": 2})]
    first = {}
    for rep in range(20):
        for i, (n_games, n_iter, kw) in enumerate(shapes):
            reqs = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(n_games)]
            got = play_games_native(reqs, 4096, n_iter, 6.6, 0.01, net, **kw).to_records()[0].tobytes()
            assert first.setdefault(i, got) == got
        if rep == 1:
            torch.cuda.synchronize()
            free_then = torch.cuda.mem_get_info()[0]
    torch.cuda.synchronize()
    assert free_then - torch.cuda.mem_get_info()[0] < 8 << 20, (free_then, torch.cuda.mem_get_info()[0])


def test_the_phases_add_up_to_the_call():
    """c4_play_phases: set-up + steady + tail + drain is the call's wall time (the first capture counted inside steady_s)."""
    import time

    import c4a0_amd
    from c4a0_amd.native import play_games_native

    net = _net(1, 32)
    reqs = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(3000)]
    play_games_native(reqs[:64], 4096, 6, 6.6, 0.01, net)
    st = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    play_games_native(reqs, 4096, 12, 6.6, 0.01, net, stats=st, resident_games=2048)
    dt = time.perf_counter() - t0
    ph = st["phases"]
    total = ph["setup_s"] + ph["steady_s"] + ph["tail_s"] + ph["drain_s"]
    assert total <= dt and dt - total < 0.05 + 0.1 * dt, (ph, dt)
    assert ph["graph_captures"] >= 1 and 0 < ph["recapture_s_inside_steady_and_tail"] < total


@pytest.mark.parametrize("host_loop", ["native", "python"])
def test_concurrent_callers_get_their_own_bytes(host_loop):
    """Four threads call play_games at once (ctypes releases the interpreter lock for the library's loop): a job captures HIP
    graphs, and another job's set-up in the middle of a capture used to fail it ("operation would make the legacy stream depend
    on a capturing stream") -- jobs now run one after another, and every caller gets exactly the bytes of a call made alone."""
    import threading

    import c4a0_amd

    net = _net(1, 32)
    shapes = [(0, 3000, 10, dict(resident_games=1024, concurrent_sessions=2)), (100000, 500, 20, {}), (200000, 2500, 8, dict(resident_games=2048)), (300000, 64, 30, {})]

    def job(first_id, n_games, n_iter, kw):
        reqs = [c4a0_amd.GameMetadata(first_id + i, 0, 0) for i in range(n_games)]
        if host_loop == "native":     # straight into c4_play_games_bf16 (not through play_games' own lock): the library's exclusion is what holds
            from c4a0_amd.native import play_games_native
            return play_games_native(reqs, 4096, n_iter, 6.6, 0.01, net, **kw).to_records()[0].tobytes()
        return c4a0_amd.play_games(reqs, 4096, n_iter, 6.6, 0.01, evaluator=net, host_loop=host_loop, **kw).to_records()[0].tobytes()

    want = [job(*s) for s in shapes]
    for _ in range(2):
        got, errs = [None] * len(shapes), []

        def run(i):
            try:
                got[i] = job(*shapes[i])
            except Exception as e:  # noqa: BLE001
                errs.append((i, repr(e)))

        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(shapes))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errs, errs
        assert got == want


@pytest.mark.parametrize("sessions", [1, 2])
def test_a_device_side_panic_ends_the_call_and_poisons_nothing(sessions):
    """A network that answers NaN (here: a NaN in the value head's output bias) makes the reference panic in UCT (utils.rs:12); the
    library's loop returns that status from its pinned probes, gives everything back, and the next call plays as if nothing had been."""
    import c4a0_amd
    from c4a0_amd._lib import C4Error
    from c4a0_amd.native import play_games_native

    good, bad = _net(1, 32), _net(1, 32)
    bad.val_b32.fill_(float("nan"))
    reqs = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(600)]
    kw = dict(resident_games=512, concurrent_sessions=sessions)
    want = play_games_native(reqs, 4096, 8, 6.6, 0.01, good, **kw).to_records()[0].tobytes()
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        with pytest.raises(C4Error, match="NAN|nan|NaN"):
            play_games_native(reqs, 4096, 8, 6.6, 0.01, bad, **kw)
    torch.cuda.synchronize()
    assert free_before - torch.cuda.mem_get_info()[0] < 8 << 20
    assert play_games_native(reqs, 4096, 8, 6.6, 0.01, good, **kw).to_records()[0].tobytes() == want


@pytest.mark.parametrize("host_loop", ["native", "python"])
def test_ctrl_c_stops_a_job_inside_the_library(host_loop):
    """The whole job is one library call; SIGINT during it must still reach the caller promptly (the Python loop is interruptible
    between replays, the reference's process is simply killed): the call runs on a helper thread, the interrupted caller asks the
    job to stop (`c4_play_games_cancel` -> C4_ERR_CANCELLED inside, KeyboardInterrupt outside), everything is given back and the
    next job plays as if nothing had been."""
    import os
    import signal
    import threading
    import time

    import c4a0_amd
    from c4a0_amd._lib import ERR_CANCELLED, lib

    net = _net(1, 32)
    small = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(300)]
    want = c4a0_amd.play_games(small, 4096, 10, 6.6, 0.01, evaluator=net).to_records()[0].tobytes()
    torch.cuda.synchronize()
    lib().c4_trim_cached_memory()                                           # (arenas are kept between calls: INTEGRATION.md "Memory between calls")
    torch.cuda.empty_cache()                                                # (and so are the Python loop's activation buffers, by torch's allocator)
    free_before = torch.cuda.mem_get_info()[0]
    big = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(200000)]           # ~ 7 s of play if left alone
    threading.Timer(0.5, lambda: os.kill(os.getpid(), signal.SIGINT)).start()
    t0 = time.perf_counter()
    with pytest.raises(KeyboardInterrupt):     # ("python": the loop of session.py, interruptible between replays -- it must give everything back too)
        c4a0_amd.play_games(big, 4096, 100, 6.6, 0.01, evaluator=net, host_loop=host_loop)
    assert time.perf_counter() - t0 < 3.0
    assert ERR_CANCELLED == 9
    lib().c4_play_games_cancel()                                            # with no job running: no effect on the next one
    assert c4a0_amd.play_games(small, 4096, 10, 6.6, 0.01, evaluator=net).to_records()[0].tobytes() == want
    torch.cuda.synchronize()
    lib().c4_trim_cached_memory()
    torch.cuda.empty_cache()
    assert free_before - torch.cuda.mem_get_info()[0] < 8 << 20


@pytest.mark.parametrize("host_loop", ["native", "python"])
def test_a_job_that_does_not_fit_fails_cleanly(host_loop):
    """Device memory nearly full (somebody else's tensors): the job's arenas cannot be allocated -- the call must say so, keep
    nothing, and the same job must run once the memory is there."""
    import c4a0_amd
    from c4a0_amd._lib import C4Error, lib

    net = _net(1, 32)
    reqs = [c4a0_amd.GameMetadata(g, 0, 0) for g in range(2048)]
    lib().c4_trim_cached_memory()
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    free = torch.cuda.mem_get_info()[0]
    hog = torch.empty(free - (1 << 30), dtype=torch.uint8, device="cuda:0")       # leaves 1 GB; the job below wants 2 048 x 43 x 900 x 128 B = 10 GB
    with pytest.raises((C4Error, RuntimeError, MemoryError)):
        c4a0_amd.play_games(reqs, 4096, 900, 6.6, 0.01, evaluator=net, host_loop=host_loop)
    torch.cuda.synchronize()
    lib().c4_trim_cached_memory()
    assert torch.cuda.mem_get_info()[0] > (1 << 30) - (64 << 20)                   # what the failed call took is back
    del hog
    torch.cuda.empty_cache()
    small = c4a0_amd.play_games(reqs[:200], 4096, 12, 6.6, 0.01, evaluator=net, host_loop=host_loop)
    assert len(small.results) == 200 and all(len(r.samples) >= 7 for r in small.results)
