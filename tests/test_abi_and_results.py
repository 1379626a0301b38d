"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol the
header declares (no compute without a GPU), the product never imports the oracle, and the
result classes mirror the reference's PyO3 classes (types.rs, pybridge.rs)."""
import ctypes as C
import os
import pickle
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from c4a0_amd import _lib

    hdr = open(os.path.join(ROOT, "include", "c4a0_hip.h")).read()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(c4_\w+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = _lib.lib()
    for name in declared:
        assert getattr(L, name) is not None


def test_struct_layouts_match_header():
    from c4a0_amd import _lib
    from c4a0_amd.session import SAMPLE_DTYPE

    assert C.sizeof(_lib.SampleRec) == 64 == SAMPLE_DTYPE.itemsize
    assert C.sizeof(_lib.GameMetadataC) == 24
    assert C.sizeof(_lib.Config) == 36          # since version 7: + reclaim_period
    assert C.sizeof(_lib.Counters) == 15 * 8 + 8   # version 7: + reclaim_passes, reclaim_blocks
    assert [n for n, *_ in _lib.SampleRec._fields_] == list(SAMPLE_DTYPE.names)


def test_no_gpu_is_reported_not_faked():
    """Without a device the product refuses to run; it never falls back to a CPU path."""
    import torch
    from c4a0_amd import _lib

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    cfg = _lib.Config(4, 0, 10, 6.6, 0.01, 0, 0, 0)
    h = C.c_void_p()
    assert _lib.lib().c4_session_create(C.byref(cfg), C.byref(h)) == _lib.ERR_NO_DEVICE
    from c4a0_amd import GameMetadata, play_games
    with pytest.raises(RuntimeError):
        play_games([GameMetadata(0, 0, 0)], 8, 2, 1.4, 0.01, lambda m, x: None)


def test_product_never_imports_the_oracle():
    for dirpath, _dirs, files in os.walk(os.path.join(ROOT, "c4a0_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "c4oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


# ---- result classes -------------------------------------------------------------------------
def _mk():
    from c4a0_amd import GameMetadata, GameResult, PlayGamesResult, Sample

    f = np.float32
    s0 = Sample(0, 0, [f(1) / f(7)] * 7, f(-0.93), f(-1.0))
    s1 = Sample(0b1000, 0, [0.5, 0.25, 0.125, 0.0625, 0.03125, 0.015625, 0.015625], f(0.93), f(1.0))
    term = Sample(0b1111 | (0b111 << 7), 0, [f(1) / f(7)] * 7, f(-0.93), f(-1.0))  # opponent has 4 on the bottom row
    g0 = GameResult(GameMetadata(7, 1, 2), [s0, s1, term])
    g1 = GameResult(GameMetadata((1 << 64) - 1, 0, 0), [s0, term])
    return PlayGamesResult([g0, g1]), (s0, s1, term)


def test_sample_to_numpy_and_flip_h():
    pgr, (s0, s1, term) = _mk()
    pos, pol, qp, qn = s1.to_numpy()                        # types.rs:125-147, training.py:329-332
    assert pos.shape == (2, 6, 7) and pos.dtype == np.float32 and pol.shape == (7,) and qp.shape == () and qn.shape == ()
    assert pos[1, 0, 3] == 1.0 and pos.sum() == 1.0         # one opponent piece at row 0 col 3
    fl = s1.flip_h()                                        # types.rs:115-122
    assert fl.mask == 0b1000 and np.array_equal(fl.policy, s1.policy[::-1])
    s = type(s1)(0b1, 0b1, s1.policy, 0.0, 0.0).flip_h()
    assert s.mask == 1 << 6 and s.value == 1 << 6
    assert s1.flip_h().flip_h() == s1
    assert s1.pos_str().splitlines()[-1] == "⚫⚫⚫🔵⚫⚫⚫"


def test_player0_score():                                   # types.rs:77-99
    pgr, _ = _mk()
    # terminal sample: 7 pieces (odd ply), OpponentWin for the side to move => player 1 to move lost => player 0 won
    assert pgr.results[0].player0_score() == 1.0
    from c4a0_amd import GameMetadata, GameResult
    with pytest.raises(RuntimeError):
        GameResult(GameMetadata(0, 0, 0), []).player0_score()


def test_cbor_wire_format_is_serde_cbor_shaped():
    """pybridge.rs:73-92: serde_cbor 0.11.2 output for the derive(Serialize) structs: maps keyed
    by field names in declaration order, shortest ints, f32 as half when lossless."""
    pgr, _ = _mk()
    b = pgr.to_cbor()
    assert b[:10] == bytes([0xA1, 0x67]) + b"results" + bytes([0x82])        # {"results": [2 items
    assert b[10:20] == bytes([0xA2, 0x68]) + b"metadata"                     # {"metadata": ...
    assert bytes([0x67]) + b"game_id" + bytes([0x07]) in b                   # game_id: 7 (1 byte)
    assert bytes([0x1B]) + b"\xff" * 8 in b                                  # u64::MAX as 8-byte uint
    assert bytes([0x66]) + b"policy" + bytes([0x87, 0xF9, 0x38, 0x00, 0xF9, 0x34, 0x00]) in b  # [0.5, 0.25 as f16
    assert bytes([0xFA]) + np.array([np.float32(1) / np.float32(7)], dtype=">f4").tobytes() in b  # 1/7 needs f32
    from c4a0_amd.results import _py_from_cbor, _py_to_cbor
    assert b == _py_to_cbor(pgr.results)                                     # the library's encoder == the per-sample Python writer
    back = type(pgr).from_cbor(b)
    assert back._lazy is not None                                            # decoded into records, no objects yet
    assert back == pgr and back.to_cbor() == b and back.results == _py_from_cbor(b)
    for bad in (b[:-3], b"\x00", b"", b + b"\x00", b[:40], b.replace(b"policy", b"pol1cy"), b.replace(bytes([0x66]) + b"policy" + bytes([0x87]), bytes([0x66]) + b"policy" + bytes([0x86]))):
        with pytest.raises(ValueError):
            type(pgr).from_cbor(bad)
        with pytest.raises(ValueError):                                      # ... and the checker refuses the same documents
            _py_from_cbor(bad)
    assert type(pgr).from_cbor(bytearray(b)) == pgr and type(pgr).from_cbor(memoryview(b)) == pgr


def test_pickle_add_unique_split():
    pgr, _ = _mk()
    assert pickle.loads(pickle.dumps(pgr)) == pgr           # pybridge.rs:83-92 __getstate__/__setstate__
    both = pgr + pgr                                        # pybridge.rs:95-106
    assert len(both.results) == 4 and len(pgr.results) == 2
    assert pgr.unique_positions() == 3                      # pybridge.rs:150-157
    ids = [r.metadata.game_id for r in both.results]
    tr1, te1 = both.split_train_test(0.5, 1337)             # pybridge_test.py:22-39
    assert [r.metadata.game_id for r in both.results] == ids
    tr2, te2 = both.split_train_test(0.5, 1337)
    assert [s.pos_str() for s in tr1] == [s.pos_str() for s in tr2] and [s.pos_str() for s in te1] == [s.pos_str() for s in te2]
    assert len(tr1) + len(te1) == sum(len(r.samples) for r in both.results)
    assert both.split_train_test(1.0, 1)[1] == [] and both.split_train_test(0.0, 1)[0] == []
    with pytest.raises(NotImplementedError):
        pgr.score_policies("a", "b", "c")


def test_split_train_test_rounds_half_away_from_zero():
    """pybridge.rs:114: `(len as f32 * train_frac).round()` -- Rust's f32::round rounds halves away
    from zero (5 games at 0.5 -> 3 training games), numpy's round would give 2."""
    from c4a0_amd import GameMetadata, GameResult, PlayGamesResult, Sample
    mk = lambda i: GameResult(GameMetadata(i, 0, 0), [Sample(1 << i, 0, [1 / 7] * 7, 0.0, 0.0)])
    for n, frac, want in [(5, 0.5, 3), (3, 0.5, 2), (7, 0.5, 4), (4, 0.5, 2), (10, 0.25, 3), (10, 0.24, 2), (1, 0.5, 1)]:
        tr, te = PlayGamesResult([mk(i) for i in range(n)]).split_train_test(frac, 7)
        assert (len(tr), len(te)) == (want, n - want), (n, frac)


def test_split_train_test_saturates_like_rust_as_usize():
    """`.round() as usize` saturates (NaN -> 0, negative -> 0, +inf -> max): no exception here either."""
    from c4a0_amd import GameMetadata, GameResult, PlayGamesResult, Sample
    mk = lambda i: GameResult(GameMetadata(i, 0, 0), [Sample(1 << i, 0, [1 / 7] * 7, 0.0, 0.0)])
    res = PlayGamesResult([mk(i) for i in range(4)])
    for frac, want in [(float("nan"), 0), (-1.0, 0), (float("-inf"), 0), (float("inf"), 4), (1e30, 4), (2.0, 4)]:
        tr, te = res.split_train_test(frac, 3)
        assert (len(tr), len(te)) == (want, 4 - want), frac


def test_results_from_records_roundtrip():
    from c4a0_amd import GameMetadata
    from c4a0_amd.results import results_from_records
    from c4a0_amd.session import SAMPLE_DTYPE

    recs = np.zeros(5, dtype=SAMPLE_DTYPE)
    recs["game_id"] = [3, 3, 3, 9, 9]
    recs["mask"] = [0, 1, 3, 0, 8]
    recs["policy"][:, 2] = 1.0
    recs["q_penalty"] = [0.5, -0.5, 0.5, 1, -1]
    out = results_from_records([GameMetadata(3, 0, 0), GameMetadata(9, 0, 0)], recs, np.array([3, 2], dtype=np.uint32))
    assert [len(r.samples) for r in out.results] == [3, 2]
    assert out.results[1].samples[1].mask == 8 and out.results[0].samples[1].q_penalty == np.float32(-0.5)


def test_install_as_c4a0_rust_makes_pickles_interchangeable():
    """The reference pickles `c4a0_rust.PlayGamesResult` with CBOR bytes as state (pybridge.rs:60,
    83-92; training.py:48-67).  A pickle written that way -- built here by hand, protocol 2
    copyreg form -- must load, and ours must carry the same module/class names."""
    import subprocess
    import sys

    code = r"""
import pickle, sys
sys.path.insert(0, %r)
import c4a0_amd
from c4a0_amd import GameMetadata, GameResult, PlayGamesResult, Sample
mod = c4a0_amd.install_as_c4a0_rust()
import c4a0_rust
assert c4a0_rust is mod and c4a0_rust.N_COLS == 7 and c4a0_rust.GameMetadata is GameMetadata
pgr = PlayGamesResult([GameResult(GameMetadata(5, 0, 0), [Sample(0, 0, [1 / 7] * 7, 0.5, 1.0)])])
blob = pickle.dumps(pgr, protocol=2)
assert b"c4a0_rust" in blob and b"PlayGamesResult" in blob
# what the reference writes: copyreg.__newobj__(c4a0_rust.PlayGamesResult) + BUILD with the CBOR bytes
ref_style = (b"\x80\x02ccopy_reg\n__newobj__\nq\x00cc4a0_rust\nPlayGamesResult\nq\x01\x85q\x02Rq\x03"
             + pickle.dumps(pgr.to_cbor(), protocol=2)[2:-1] + b"b.")
back = pickle.loads(ref_style)
assert type(back) is PlayGamesResult and back == pgr
assert pickle.loads(blob) == pgr
print("ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-2000:]


def test_lazy_results_and_bulk_arrays_agree_with_objects():
    from c4a0_amd import GameMetadata
    from c4a0_amd.results import PlayGamesResult, results_from_records
    from c4a0_amd.session import SAMPLE_DTYPE

    rng = np.random.default_rng(0)
    counts = np.array([3, 0, 5, 2], dtype=np.uint32)
    n = int(counts.sum())
    recs = np.zeros(n, dtype=SAMPLE_DTYPE)
    recs["mask"] = rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["value"] = recs["mask"] & rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["policy"] = rng.random((n, 7), dtype=np.float32)
    recs["q_penalty"] = rng.random(n, dtype=np.float32)
    recs["q_no_penalty"] = -recs["q_penalty"]
    reqs = [GameMetadata(10 + i, 0, 0) for i in range(4)]
    lazy = results_from_records(reqs, recs, counts)
    assert lazy._lazy is not None
    planes, pol, qp, qn, gidx = lazy.to_arrays()                     # no objects built
    assert lazy._lazy is not None and lazy.unique_positions() == len({(int(m), int(v)) for m, v in zip(recs["mask"], recs["value"])})
    assert planes.shape == (n, 2, 6, 7) and np.array_equal(gidx, [0, 0, 0, 2, 2, 2, 2, 2, 3, 3])
    objs = lazy.results                                              # now materialised
    assert lazy._lazy is None and [len(r.samples) for r in objs] == [3, 0, 5, 2]
    flat = [s for r in objs for s in r.samples]
    for i, s in enumerate(flat):
        p, po, a, b = s.to_numpy()
        assert np.array_equal(p, planes[i]) and np.array_equal(po, pol[i]) and a == qp[i] and b == qn[i]
    planes2, pol2, qp2, qn2, gidx2 = lazy.to_arrays()                 # same answer from the object path
    assert np.array_equal(planes, planes2) and np.array_equal(pol, pol2) and np.array_equal(gidx, gidx2)
    assert PlayGamesResult.from_cbor(lazy.to_cbor()) == lazy


def test_cbor_round_trip_property():
    """Property test (the reference uses proptest for its own invariants): any PlayGamesResult --
    ids across the whole u64 range, bitboards, f32 values incl. half-representable ones, signed
    zeros, subnormals and infinities -- survives to_cbor/from_cbor bit for bit, re-encodes to the
    same bytes, and concatenation commutes with encoding of the parts' results."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from c4a0_amd.results import GameMetadata, GameResult, PlayGamesResult, Sample, _py_from_cbor, _py_to_cbor

    # every unsigned width serde_cbor packs to (1, 2, 3, 5, 9 bytes) and every float class (half normal / subnormal / boundary, f32 only)
    u64 = st.one_of(st.integers(0, (1 << 64) - 1), st.sampled_from([0, 23, 24, 255, 256, 65535, 65536, (1 << 32) - 1, 1 << 32, (1 << 64) - 1]))
    f32 = st.one_of(st.floats(width=32, allow_nan=False), st.floats(width=16, allow_nan=False),
                    st.sampled_from([0.5, 0.25, -0.0, 1.0 / 7.0, 65504.0, 65520.0, 65505.0, 2.0 ** -24, 2.0 ** -25, 3 * 2.0 ** -24, 2.0 ** -14, 1023 * 2.0 ** -24,
                                     2.0 ** -14 + 2.0 ** -25, 1e-45, float("inf"), float("-inf")]))

    @st.composite
    def samples(draw):
        mask = draw(st.one_of(st.integers(0, (1 << 42) - 1), u64))
        value = draw(st.integers(0, (1 << 64) - 1)) & mask
        pol = np.array(draw(st.lists(f32, min_size=7, max_size=7)), dtype=np.float32)
        return Sample(mask, value, pol, np.float32(draw(f32)), np.float32(draw(f32)))

    games = st.builds(lambda a, b, c, ss: GameResult(GameMetadata(a, b, c), ss), u64, u64, u64, st.lists(samples(), max_size=4))

    @settings(max_examples=120, deadline=None)
    @given(st.lists(games, max_size=4), st.lists(games, max_size=3))
    def check(g1, g2):
        a, b = PlayGamesResult(g1), PlayGamesResult(g2)
        for r in (a, b, a + b):
            enc = r.to_cbor()
            assert enc == _py_to_cbor(r.results)            # native encoder (on records) == per-sample Python writer
            back = PlayGamesResult.from_cbor(enc)
            assert back.to_cbor() == enc and back.results == _py_from_cbor(enc)
            assert len(back.results) == len(r.results)
            for x, y in zip(back.results, r.results):
                assert (x.metadata.game_id, x.metadata.player0_id, x.metadata.player1_id) == \
                       (y.metadata.game_id, y.metadata.player0_id, y.metadata.player1_id)
                assert len(x.samples) == len(y.samples)
                for s, t in zip(x.samples, y.samples):
                    assert (s.mask, s.value) == (t.mask, t.value)
                    assert np.asarray(s.policy, np.float32).tobytes() == np.asarray(t.policy, np.float32).tobytes()
                    assert np.float32(s.q_penalty).tobytes() == np.float32(t.q_penalty).tobytes()
                    assert np.float32(s.q_no_penalty).tobytes() == np.float32(t.q_no_penalty).tobytes()

    check()


def test_native_decoder_accepts_what_the_python_checker_accepts():
    """Where an f32 is expected serde's visitor takes any number: halves, singles, doubles and integers.  A document written with
    those widths by hand decodes to the same records through the library and through the Python checker, NaN payloads collapse to
    the quiet NaN serde_cbor writes (f9 7e00), and the record bookkeeping (game_id copy, index | terminal flag) is the generator's."""
    import struct
    from c4a0_amd.results import PlayGamesResult, _K, _cbor_uint, _py_from_cbor

    def sample(mask, value, floats):
        assert len(floats) == 9
        return b"".join([_cbor_uint(5, 4), _K["pos"], _cbor_uint(5, 2), _K["mask"], _cbor_uint(0, mask), _K["value"], _cbor_uint(0, value),
                         _K["policy"], _cbor_uint(4, 7)] + floats[:7] + [_K["q_penalty"], floats[7], _K["q_no_penalty"], floats[8]])

    f16 = lambda x: b"\xf9" + struct.pack(">e", x)
    f32 = lambda x: b"\xfa" + struct.pack(">f", x)
    f64 = lambda x: b"\xfb" + struct.pack(">d", x)
    fl = [f16(0.5), f32(0.1), f64(0.1), _cbor_uint(0, 3), _cbor_uint(0, 1 << 40), f16(6e-8), f64(1e300), b"\xf9\x7e\x01", f32(float("-inf"))]
    doc = b"".join([_cbor_uint(5, 1), _K["results"], _cbor_uint(4, 1), _cbor_uint(5, 2), _K["metadata"], _cbor_uint(5, 3),
                    _K["game_id"], _cbor_uint(0, 77), _K["player0_id"], _cbor_uint(0, 1), _K["player1_id"], _cbor_uint(0, 2),
                    _K["samples"], _cbor_uint(4, 2), sample(5, 1, fl), sample(7, 2, fl[::-1])])
    got = PlayGamesResult.from_cbor(doc)
    recs, counts = got.to_records()
    assert counts.tolist() == [2] and recs["game_id"].tolist() == [77, 77] and recs["meta"].tolist() == [0, 1 | (1 << 16)]
    want = _py_from_cbor(doc)
    with np.errstate(over="ignore"):
        for s, t in zip(got.results[0].samples, want[0].samples):
            assert (s.mask, s.value) == (t.mask, t.value)
            assert np.array_equal(s.policy.view(np.uint32) & 0xFFC00000, t.policy.view(np.uint32) & 0xFFC00000)   # NaN payload bits aside
            assert np.array_equal(s.policy[~np.isnan(s.policy)], t.policy[~np.isnan(t.policy)])
    assert np.isinf(recs["policy"][0][6]) and recs["policy"][0][3] == 3.0 and recs["policy"][0][4] == np.float32(1 << 40)
    assert np.isnan(recs["q_penalty"][0]) and recs["q_no_penalty"][0] == -np.inf
    # NaN / infinities are written as serde_cbor writes them
    enc = got.to_cbor()
    assert b"\xf9\x7e\x00" in enc and b"\xf9\xfc\x00" in enc and b"\xf9\x7c\x00" in enc


def test_codec_throughput_keeps_up_with_the_generator():
    """the generator emits ~0.5 M samples per second of play; pickling a result (src/c4a0/training.py:62-63:
    `pickle.dump(games, f)` = __getstate__ = to_cbor) must not be slower than playing it.  Bound: >= 1 M samples/s each way on
    whatever core runs this test (the library does 5-30 M; the per-sample Python writer it replaced did 0.025 M)."""
    import time
    from c4a0_amd.results import PlayGamesResult, results_from_records
    from c4a0_amd.session import SAMPLE_DTYPE

    rng = np.random.default_rng(5)
    g = 4096
    counts = rng.integers(8, 30, g).astype(np.uint32)
    n = int(counts.sum())
    recs = np.zeros(n, dtype=SAMPLE_DTYPE)
    recs["mask"] = rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["value"] = recs["mask"] & rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["policy"] = rng.integers(0, 100, (n, 7)).astype(np.float32) / np.float32(100)
    recs["q_penalty"] = rng.random(n, dtype=np.float32)
    recs["q_no_penalty"] = np.sign(recs["q_penalty"])
    ids = np.stack([np.arange(g, dtype=np.uint64), np.zeros(g, np.uint64), np.ones(g, np.uint64)], 1)
    res = results_from_records(ids, recs, counts)
    res.to_cbor()                                           # first call: library load
    t0 = time.perf_counter()
    blob = pickle.dumps(res)
    t1 = time.perf_counter()
    back = pickle.loads(blob)
    t2 = time.perf_counter()
    assert back == res and back._lazy is not None and res._lazy is not None
    assert n / (t1 - t0) > 1e6 and n / (t2 - t1) > 1e6, (n, t1 - t0, t2 - t1)
    assert (res + back).to_cbor() == PlayGamesResult.from_cbor(res.to_cbor()).__add__(back).to_cbor()   # record-form concatenation
    assert len(res + back) == 2 * g


def test_split_train_test_is_rands_slice_shuffle():
    """pybridge.rs:110-116: `results.shuffle(&mut StdRng::seed_from_u64(seed))`, then the first round(len * frac) games train.  The
    library's permutation (c4_shuffle_games) equals the oracle's restatement, which the rand crate's own vectors pin
    (tests/test_oracle_libm_rng.py); lazy and object-built results split identically and `self` keeps its order and its form."""
    from c4a0_amd.results import GameMetadata, GameResult, PlayGamesResult, Sample, results_from_records, shuffled_game_order
    from c4a0_amd.session import SAMPLE_DTYPE
    from oracle import c4oracle as O

    for seed in (0, 1, 1337, (1 << 64) - 1, 0xDEADBEEFCAFE):
        for n in (0, 1, 2, 3, 12, 13, 14, 100, 1000, 4097, 70001):
            got = shuffled_game_order(n, seed)
            assert np.array_equal(got, O.shuffle_games(seed, n).astype(np.int64)), (seed, n)
    rng = np.random.default_rng(1)
    counts = rng.integers(0, 5, 37).astype(np.uint32)
    recs = np.zeros(int(counts.sum()), dtype=SAMPLE_DTYPE)
    recs["mask"] = np.repeat(np.arange(37, dtype=np.uint64), counts.astype(np.int64))      # every sample carries its game's index
    recs["policy"] = rng.random((len(recs), 7), dtype=np.float32)
    ids = np.stack([np.arange(37, dtype=np.uint64)] * 3, 1)
    lazy = results_from_records(ids, recs, counts)
    objs = PlayGamesResult(results_from_records(ids, recs, counts).results)
    for frac, seed in ((0.5, 1337), (0.8, 0), (0.0, 5), (1.0, 5)):
        order = O.shuffle_games(seed, 37).tolist()
        n_train = int(np.floor(float(np.float32(37) * np.float32(frac)) + 0.5))
        want_train = [g for g in order[:n_train] for _ in range(counts[g])]
        want_test = [g for g in order[n_train:] for _ in range(counts[g])]
        for res in (lazy, objs):
            tr, te = res.split_train_test(frac, seed)
            assert [s.mask for s in tr] == want_train and [s.mask for s in te] == want_test
        a, b = lazy.split_train_test(frac, seed), objs.split_train_test(frac, seed)
        assert a[0] == b[0] and a[1] == b[1]
    assert lazy._lazy is not None and [r.metadata.game_id for r in objs.results] == list(range(37))
    # negative seeds never reach the reference (PyO3 extracts a u64); the low 64 bits are used here
    assert np.array_equal(shuffled_game_order(20, -1), shuffled_game_order(20, (1 << 64) - 1))


def test_merge_parts_restores_request_order_numpy_and_torch():
    """c4a0_amd.api.merge_parts (used for concurrent sessions and for the multi-GPU shard merge): parts
    that own interleaved request positions come back in request order, numpy records and torch uint8
    rows alike, including empty parts and games without samples."""
    import torch
    from c4a0_amd.api import merge_parts
    from c4a0_amd.session import SAMPLE_DTYPE

    rng = np.random.default_rng(3)
    n = 53
    counts = rng.integers(0, 6, size=n).astype(np.uint32)
    recs = np.zeros(int(counts.sum()), dtype=SAMPLE_DTYPE)
    recs["game_id"] = np.repeat(np.arange(n, dtype=np.uint64), counts)
    recs["meta"] = np.concatenate([np.arange(c, dtype=np.uint32) for c in counts]) if counts.sum() else []
    recs["mask"] = rng.integers(0, 1 << 42, size=len(recs)).astype(np.uint64)
    offs = np.concatenate([[0], np.cumsum(counts.astype(np.int64))])
    for k in (1, 2, 3, 8, 60):     # 60 > n: some parts own nothing
        parts_np, parts_t = [], []
        for p in range(k):
            pos = np.arange(p, n, k, dtype=np.int64)
            r = np.concatenate([recs[offs[g]:offs[g + 1]] for g in pos]) if len(pos) else recs[:0]
            parts_np.append((pos, counts[pos], r))
            parts_t.append((pos, counts[pos], torch.from_numpy(r.view(np.uint8).reshape(-1, 64).copy())))
        got, got_counts = merge_parts(n, parts_np)
        assert np.array_equal(got_counts, counts) and got.tobytes() == recs.tobytes()
        got_t, got_counts_t = merge_parts(n, parts_t)
        assert np.array_equal(got_counts_t, counts) and got_t.numpy().tobytes() == recs.tobytes()


def test_to_records_round_trip():
    from c4a0_amd.results import results_from_records
    pgr, _ = _mk()
    recs, counts = pgr.to_records()
    assert counts.tolist() == [len(r.samples) for r in pgr.results] and len(recs) == counts.sum()
    back = results_from_records([r.metadata for r in pgr.results], recs, counts)
    assert back == pgr
    r2, c2 = back.to_records()      # the lazy form hands the arrays straight back
    assert r2.tobytes() == recs.tobytes() and np.array_equal(c2, counts)


def test_tile_choice_of_a_session_alone_on_the_device():
    """InferenceNet._alone_config (host logic, no GPU): only a session that has the device to itself asks for tiles by number --
    up to 1 024 rows the wave-specialised small tiles (41 / 42 / 44 / 43: faster alone, 5 % slower beside a second session, round 5),
    between 1 025 and 1 728 rows the 96 x 96 tile with loader wavefronts (the 2F-wide layer: 128 x 96 / 192 x 96), beyond that the wave-specialised 128 x 96 / 128 x 192 tiles;
    otherwise the library's automatic choice (0) stands -- round 3's small tiles up to 1 024 rows, config 11 above."""
    from c4a0_amd.nn import InferenceNet

    net = object.__new__(InferenceNet)
    net.latency_mode = False
    assert [net._alone_config(m, 2688, 1344) for m in (1, 512, 1024, 1025, 1700, 4096)] == [0] * 6
    assert net._alone_config(512, 1344, 1344, latency=True) == 41          # the per-call answer (forward_numpy) overrides the attribute
    net.latency_mode = True
    assert [net._alone_config(m, 2688, 1344) for m in (256, 384, 385, 576, 577, 864, 865, 1024)] == [41, 41, 42, 42, 44, 44, 43, 43]
    assert [net._alone_config(m, 1344, 1344) for m in (256, 512, 513, 768, 1024)] == [41, 41, 42, 42, 42]
    # 1 025 - 1 728 rows: the F-wide layers on 96 x 96 (4 + 4 wavefronts); the 2F-wide layer on 128 x 96 while 9 x 28 tiles are one wave of
    # workgroups, then on 192 x 96 (round 5: a quarter fewer operand bytes per CU)
    assert net._alone_config(1025, 1344, 1344) == 44 and net._alone_config(1728, 1344, 1344) == 44
    assert [net._alone_config(m, 2688, 1344) for m in (1025, 1152, 1153, 1728)] == [43, 43, 59, 59]
    net.wide_tiles_r5 = False
    assert net._alone_config(1025, 2688, 1344) == 44 and net._alone_config(1728, 2688, 1344) == 44
    net.wide_tiles_r5 = True
    assert net._alone_config(1729, 2688, 1344) == 35 and net._alone_config(1729, 1344, 1344) == 43
    # the 64-channel net (k = 2 688): the automatic choice for the 2F-wide layer, the 128 x 192 tile for the F-wide ones above 1 024 rows
    assert net._alone_config(4096, 2688, 2688) == 11 and net._alone_config(1500, 2688, 2688) == 11
    assert net._alone_config(4096, 5376, 2688) == 0 and net._alone_config(1500, 5376, 2688) == 0 and net._alone_config(512, 2688, 2688) == 0
    net.use_loader_waves = False                                            # the A/B switch: no loader-wavefront form anywhere
    assert [net._alone_config(m, 2688, 1344) for m in (512, 1500, 4096)] == [0, 23, 11]


def test_gemm_block_to_tile_map_is_a_bijection_for_every_grid():
    """c4_linear_bf16's block -> tile map (XCD rectangles, idx / rn by multiplication) walked on the host with the kernels' own
    formula.  ADVICE r4: an XCD rectangle one tile wide (rn == 1; N in {192, 384, 768, 1536}) sent every block but the first
    past N.  Every (tile shape, m, n) the entry point accepts must map its blocks one-to-one onto the tile grid."""
    import ctypes as C
    from c4a0_amd import _lib
    L = _lib.lib()
    shapes = [(128, 192), (256, 192), (64, 192), (128, 96), (64, 96), (96, 96), (64, 64), (192, 192), (96, 64), (192, 96), (96, 192)]   # the last two: configs 54-59 (round 5)
    for bm, bn in shapes:
        for n in [192 * i for i in (1, 2, 3, 4, 7, 8, 14, 16, 28)] + [2688 * 2]:
            if n % bn:
                continue
            for m in (1, bm, 2 * bm, 3 * bm + 1, 8 * bm, 2048, 1700, 4096, 8 * bm * 5):
                nb = C.c_uint32(0)
                _lib.check(L.c4_linear_bf16_tile_map(m, n, bm, bn, None, 0, C.byref(nb)))
                tiles_m, tiles_n = -(-m // bm), n // bn
                assert nb.value == tiles_m * tiles_n
                out = (C.c_uint32 * (2 * nb.value))()
                _lib.check(L.c4_linear_bf16_tile_map(m, n, bm, bn, out, nb.value, C.byref(nb)))
                seen = {(out[2 * b], out[2 * b + 1]) for b in range(nb.value)}
                assert seen == {(i, j) for i in range(tiles_m) for j in range(tiles_n)}, (bm, bn, m, n)
    # more tiles along m than the packed 16-bit kernel argument holds: refused, not wrapped
    nb = C.c_uint32(0)
    assert L.c4_linear_bf16_tile_map(64 * 70001, 192, 64, 64, None, 0, C.byref(nb)) != 0


def test_request_table_and_resident_games_defaults():
    """Host logic of play_games that needs no GPU: the request list is read in ONE pass into a uint64[n, 3] table (bad objects ->
    TypeError as extract() fails in the reference, pybridge.rs:30; ids outside u64 -> OverflowError as PyO3's), and the number of
    resident games when the caller does not say: everything up to 4 096 games, then the largest of 4 096 / 8 192 / 16 384 that the
    job fills four times over (profiles/r06_whole_call.txt), 4 096 for host-callback modes."""
    from c4a0_amd import GameMetadata
    from c4a0_amd.api import _ids_of, _validate, default_resident_games

    ids = _ids_of([GameMetadata(5, 1, 2), GameMetadata((1 << 64) - 1, 0, 0)])
    assert ids.dtype == np.uint64 and ids.tolist() == [[5, 1, 2], [(1 << 64) - 1, 0, 0]]
    assert _ids_of([]).shape == (0, 3)
    with pytest.raises(TypeError):
        _ids_of([(1, 2, 3)])
    class Bad:
        game_id, player0_id, player1_id = -1, 0, 0
    with pytest.raises(OverflowError):
        _ids_of([Bad()])
    with pytest.raises(TypeError):      # two different players need one evaluator per model
        _validate([GameMetadata(0, 1, 2)], 8, 10, None, lambda x: x)
    with pytest.raises(KeyError):
        _validate([GameMetadata(0, 1, 2)], 8, 10, None, {1: None})
    got = {n: default_resident_games(n, 100, True) for n in (1, 100, 4096, 4097, 16384, 32767, 32768, 40960, 65535, 65536, 10 ** 6)}
    assert got == {1: 1, 100: 100, 4096: 4096, 4097: 4096, 16384: 4096, 32767: 4096, 32768: 8192, 40960: 8192, 65535: 8192, 65536: 16384, 10 ** 6: 16384}
    assert default_resident_games(10 ** 6, 100, False) == 4096


def test_ctypes_structures_match_the_headers_layout_as_gcc_sees_it(tmp_path):
    """Every structure the ctypes binding hands to / reads from the library has the size and the field offsets gcc computes from
    include/c4a0_hip.h (a consumer written in another language mirrors the same layout: INTEGRATION.md)."""
    import shutil
    import subprocess

    from c4a0_amd import _lib

    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("c4_game_metadata", _lib.GameMetadataC), ("c4_sample_rec", _lib.SampleRec), ("c4_config", _lib.Config), ("c4_counters", _lib.Counters),
             ("c4_network_bf16", _lib.NetworkBf16), ("c4_play_options", _lib.PlayOptions), ("c4_play_phases", _lib.PlayPhases)]
    lines = []
    for cname, cls in pairs:
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _t in cls._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "c4a0_hip.h"\nint main(void) {\n' + "\n".join(lines) + "\nreturn 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.strip().splitlines()
    for (cname, cls), line in zip(pairs, out):
        f = line.split()
        assert f[0] == cname and int(f[1]) == C.sizeof(cls), (cname, f[1], C.sizeof(cls))
        assert [int(x) for x in f[2:]] == [getattr(cls, n).offset for n, _t in cls._fields_], cname


def test_split_train_test_tensors_equal_the_per_sample_path():
    """c4a0_amd.dataset.split_train_test_tensors = what the reference builds sample by sample (training.py:207 `split_train_test`, then
    `SampleDataModule`: every sample AND its `flip_h()` through `to_numpy`, training.py:317-333) as whole tensors: same partition, same
    order, same numbers."""
    import torch
    from c4a0_amd.dataset import split_train_test_tensors
    from c4a0_amd.results import results_from_records
    from c4a0_amd.session import SAMPLE_DTYPE

    rng = np.random.default_rng(11)
    counts = rng.integers(1, 6, 23).astype(np.uint32)
    n = int(counts.sum())
    recs = np.zeros(n, dtype=SAMPLE_DTYPE)
    recs["mask"] = rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["value"] = recs["mask"] & rng.integers(0, 1 << 42, n, dtype=np.uint64)
    recs["policy"] = rng.random((n, 7), dtype=np.float32)
    recs["q_penalty"] = rng.random(n, dtype=np.float32) * 2 - 1
    recs["q_no_penalty"] = np.sign(recs["q_penalty"])
    ids = np.stack([np.arange(23, dtype=np.uint64)] * 3, 1)
    res = results_from_records(ids, recs, counts)
    for frac, seed in ((0.8, 1337), (0.5, 0), (1.0, 3), (0.0, 3)):
        train, test = res.split_train_test(frac, seed)
        got = split_train_test_tensors(res, frac, seed, device="cpu")
        for samples, tensors in ((train, got[0]), (test, got[1])):
            samples = samples + [s.flip_h() for s in samples]          # SampleDataModule.__init__
            assert tensors[0].shape[0] == len(samples)
            if not samples:
                continue
            want = [np.stack(x) for x in zip(*[s.to_numpy() for s in samples])]
            for t, w in zip(tensors, want):
                assert t.dtype == torch.float32 and np.array_equal(t.numpy(), w)
    assert res._lazy is not None
