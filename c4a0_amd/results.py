"""Result types of `play_games` -- the Python surface of the reference's PyO3 classes.

Mirrors (names, arguments, behaviour) the classes the reference registers in
rust/src/lib.rs:32-35: `GameMetadata` (types.rs:37-60), `Sample` (types.rs:103-153),
`GameResult` (types.rs:63-100) and `PlayGamesResult` (pybridge.rs:58-158), so that the
reference's callers (src/c4a0/training.py:179-207,317-333; tournament.py:84-139) work on
GPU-generated games unchanged.  Host-side bookkeeping: a result stays the packed 64-byte records
the GPU handed over (and a games x 3 table of ids) for as long as nobody asks for Python objects;
the wire format (`to_cbor` / `from_cbor` / pickling) and the train / test permutation run in the
library's host functions (include/c4a0_hip.h c4_records_to_cbor, c4_cbor_to_records,
c4_shuffle_games) on those arrays.  The pure-Python codec further down (`_py_to_cbor`,
`_py_from_cbor`) is what the tests check the native one against; the product does not call it.
"""
from __future__ import annotations

import struct
from typing import Iterable, List, Sequence, Tuple

import math

import numpy as np

N_COLS, N_ROWS = 7, 6
_COL0 = 0x810204081  # bits 0,7,..,35: column 0
_START03 = _COL0 | (_COL0 << 1) | (_COL0 << 2) | (_COL0 << 3)


def _has_four(x: int) -> bool:
    h = x & (x >> 1) & (x >> 2) & (x >> 3) & _START03
    v = x & (x >> 7) & (x >> 14) & (x >> 21)
    d1 = x & (x >> 8) & (x >> 16) & (x >> 24) & _START03
    d2 = (x >> 3) & (x >> 9) & (x >> 15) & (x >> 21) & _START03
    return (h | v | d1 | d2) != 0


def terminal_state(mask: int, value: int) -> int:
    """c4r.rs:228-238: 0 none, 1 PlayerWin, 2 OpponentWin, 3 Draw."""
    if _has_four(mask & value):
        return 1
    if _has_four(mask & ~value):
        return 2
    if bin(mask).count("1") == 42:
        return 3
    return 0


_REV7 = [int(f"{i:07b}"[::-1], 2) for i in range(128)]   # a row's 7 cells mirrored


def flip_h_bits(x: int) -> int:
    """Mirror the 7 columns of every row (c4r.rs:289-299)."""
    r = _REV7
    return (r[x & 127] | r[(x >> 7) & 127] << 7 | r[(x >> 14) & 127] << 14 | r[(x >> 21) & 127] << 21
            | r[(x >> 28) & 127] << 28 | r[(x >> 35) & 127] << 35)


class GameMetadata:
    """types.rs:37-60."""

    __slots__ = ("game_id", "player0_id", "player1_id")

    def __init__(self, game_id: int = 0, player0_id: int = 0, player1_id: int = 0):
        for v in (game_id, player0_id, player1_id):
            if not (0 <= int(v) < 1 << 64):
                raise OverflowError("GameMetadata fields are u64")
        self.game_id, self.player0_id, self.player1_id = int(game_id), int(player0_id), int(player1_id)

    def __repr__(self):
        return f"GameMetadata(game_id={self.game_id}, player0_id={self.player0_id}, player1_id={self.player1_id})"

    def __eq__(self, o):
        return isinstance(o, GameMetadata) and (self.game_id, self.player0_id, self.player1_id) == (o.game_id, o.player0_id, o.player1_id)


class Sample:
    """types.rs:103-153.  `pos` is the (mask, value) bitboard pair of c4r.rs:13-17."""

    __slots__ = ("mask", "value", "policy", "q_penalty", "q_no_penalty")

    def __init__(self, mask: int, value: int, policy: Sequence[float], q_penalty: float, q_no_penalty: float):
        self.mask, self.value = int(mask), int(value)
        self.policy = np.asarray(policy, dtype=np.float32).reshape(7).copy()
        self.q_penalty = np.float32(q_penalty)
        self.q_no_penalty = np.float32(q_no_penalty)

    @classmethod
    def _bulk(cls, recs: np.ndarray) -> List["Sample"]:
        """One Sample per record, built from whole-column conversions (no per-sample numpy field access)."""
        new = cls.__new__
        pol = np.ascontiguousarray(recs["policy"])          # one copy; each sample's policy is its row (a view of this block)
        out = []
        for m, v, p, a, b in zip(recs["mask"].tolist(), recs["value"].tolist(), pol, list(recs["q_penalty"]), list(recs["q_no_penalty"])):
            s = new(cls)
            s.mask, s.value, s.policy, s.q_penalty, s.q_no_penalty = m, v, p, a, b
            out.append(s)
        return out

    def flip_h(self) -> "Sample":  # types.rs:115-122
        s = Sample.__new__(Sample)
        s.mask, s.value, s.policy = flip_h_bits(self.mask), flip_h_bits(self.value), self.policy[::-1].copy()
        s.q_penalty, s.q_no_penalty = self.q_penalty, self.q_no_penalty
        return s

    def to_numpy(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:  # types.rs:125-147
        # c4r.rs:378-392: plane 0 = the bits of `value` (the side to move), plane 1 = the bits of `mask & !value`, bit = row * 7 + col
        planes = np.unpackbits(np.array([self.value, self.mask & ~self.value], dtype="<u8").view(np.uint8), bitorder="little")
        pos = planes.reshape(2, 64)[:, :42].astype(np.float32).reshape(2, N_ROWS, N_COLS)
        return pos, self.policy.copy(), np.array(self.q_penalty, dtype=np.float32), np.array(self.q_no_penalty, dtype=np.float32)

    def pos_str(self) -> str:  # types.rs:150-152 / c4r.rs:395-413
        rows = []
        for row in range(N_ROWS - 1, -1, -1):
            s = ""
            for col in range(N_COLS):
                bit = 1 << (row * 7 + col)
                s += "⚫" if not self.mask & bit else ("🔴" if self.value & bit else "🔵")
            rows.append(s)
        return "\n".join(rows)

    def __eq__(self, o):
        return (isinstance(o, Sample) and (self.mask, self.value) == (o.mask, o.value)
                and self.policy.tobytes() == o.policy.tobytes()
                and self.q_penalty.tobytes() == o.q_penalty.tobytes() and self.q_no_penalty.tobytes() == o.q_no_penalty.tobytes())

    def __repr__(self):
        return f"Sample(mask={self.mask:#x}, value={self.value:#x}, policy={self.policy.tolist()}, q_penalty={float(self.q_penalty)}, q_no_penalty={float(self.q_no_penalty)})"


class GameResult:
    """types.rs:63-100."""

    __slots__ = ("metadata", "samples")

    def __init__(self, metadata: GameMetadata, samples: List[Sample]):
        self.metadata, self.samples = metadata, samples

    def player0_score(self) -> float:  # types.rs:77-99
        for s in self.samples:
            t = terminal_state(s.mask, s.value)
            if t:
                score = {1: 1.0, 2: 0.0, 3: 0.5}[t]
                return 1.0 - score if bin(s.mask).count("1") % 2 == 1 else score
        raise RuntimeError("player0_score called on an unfinished game")  # reference panics

    def __eq__(self, o):
        return isinstance(o, GameResult) and self.metadata == o.metadata and self.samples == o.samples


# ---------------------------------------------------------------------------------------------
# CBOR wire format of PlayGamesResult (pybridge.rs:73-92): what serde_cbor 0.11.2 emits for the
# derive(Serialize) structs -- definite-length maps keyed by field name in declaration order,
# unsigned ints in the shortest form, f32 as half precision when that is lossless.
# ---------------------------------------------------------------------------------------------
def _cbor_uint(major: int, n: int) -> bytes:
    if n < 24:
        return bytes([major << 5 | n])
    if n < 1 << 8:
        return bytes([major << 5 | 24, n])
    if n < 1 << 16:
        return bytes([major << 5 | 25]) + struct.pack(">H", n)
    if n < 1 << 32:
        return bytes([major << 5 | 26]) + struct.pack(">I", n)
    return bytes([major << 5 | 27]) + struct.pack(">Q", n)


def _cbor_text(s: str) -> bytes:
    b = s.encode()
    return _cbor_uint(3, len(b)) + b


def _cbor_f32(x: np.float32) -> bytes:
    x = np.float32(x)
    if np.isnan(x):
        return b"\xf9\x7e\x00"
    if np.isinf(x):
        return b"\xf9\x7c\x00" if x > 0 else b"\xf9\xfc\x00"
    with np.errstate(over="ignore"):
        h = np.float16(x)
    if np.float32(h) == x:
        return b"\xf9" + struct.pack(">e", float(h))
    return b"\xfa" + struct.pack(">f", float(x))


_K = {k: _cbor_text(k) for k in ("results", "metadata", "samples", "game_id", "player0_id", "player1_id",
                                 "pos", "policy", "q_penalty", "q_no_penalty", "mask", "value")}


class _CborReader:
    def __init__(self, data: bytes):
        self.d, self.i = memoryview(data), 0

    def head(self) -> Tuple[int, int]:
        b = self.d[self.i]
        self.i += 1
        major, info = b >> 5, b & 31
        if info < 24:
            return major, info
        n = {24: 1, 25: 2, 26: 4, 27: 8}.get(info)
        if n is None:
            raise ValueError("unsupported CBOR additional info (indefinite lengths are not produced by serde_cbor here)")
        v = int.from_bytes(self.d[self.i:self.i + n], "big")
        self.i += n
        return major, (info << 64) | v if major == 7 else v

    def uint(self) -> int:
        major, v = self.head()
        if major != 0:
            raise ValueError("expected unsigned integer")
        return v

    def length(self, want_major: int) -> int:
        major, v = self.head()
        if major != want_major:
            raise ValueError(f"expected CBOR major type {want_major}, got {major}")
        return v

    def text(self) -> str:
        n = self.length(3)
        s = bytes(self.d[self.i:self.i + n]).decode()
        self.i += n
        return s

    def f32(self) -> np.float32:
        major, v = self.head()
        if major == 7:
            info, raw = v >> 64, v & ((1 << 64) - 1)
            if info == 25:
                return np.float32(struct.unpack(">e", raw.to_bytes(2, "big"))[0])
            if info == 26:
                return np.float32(struct.unpack(">f", raw.to_bytes(4, "big"))[0])
            if info == 27:
                return np.float32(struct.unpack(">d", raw.to_bytes(8, "big"))[0])
        if major == 0:
            return np.float32(v)
        raise ValueError("expected a float")

    def struct_fields(self, names: Sequence[str]):
        n = self.length(5)
        if n != len(names):
            raise ValueError(f"expected a map of {len(names)} fields")
        for want in names:
            got = self.text()
            if got != want:
                raise ValueError(f"expected field {want!r}, got {got!r}")
            yield want


def _py_to_cbor(results: Sequence[GameResult]) -> bytes:
    """The wire format written sample by sample in Python: the CHECKER of the library's encoder (tests), not a product path."""
    out = [_cbor_uint(5, 1), _K["results"], _cbor_uint(4, len(results))]
    for r in results:
        m = r.metadata
        out += [_cbor_uint(5, 2), _K["metadata"], _cbor_uint(5, 3),
                _K["game_id"], _cbor_uint(0, m.game_id), _K["player0_id"], _cbor_uint(0, m.player0_id),
                _K["player1_id"], _cbor_uint(0, m.player1_id),
                _K["samples"], _cbor_uint(4, len(r.samples))]
        for s in r.samples:
            out += [_cbor_uint(5, 4), _K["pos"], _cbor_uint(5, 2), _K["mask"], _cbor_uint(0, s.mask),
                    _K["value"], _cbor_uint(0, s.value), _K["policy"], _cbor_uint(4, 7)]
            out += [_cbor_f32(p) for p in s.policy]
            out += [_K["q_penalty"], _cbor_f32(s.q_penalty), _K["q_no_penalty"], _cbor_f32(s.q_no_penalty)]
    return b"".join(out)


def _py_from_cbor(cbor: bytes) -> List[GameResult]:
    """The checker of the library's decoder (tests), not a product path."""
    try:
        r = _CborReader(cbor)
        results = []
        for _ in r.struct_fields(["results"]):
            for _g in range(r.length(4)):
                meta, samples = None, []
                for f in r.struct_fields(["metadata", "samples"]):
                    if f == "metadata":
                        vals = [r.uint() for _ in r.struct_fields(["game_id", "player0_id", "player1_id"])]
                        meta = GameMetadata(*vals)
                    else:
                        for _s in range(r.length(4)):
                            d = {}
                            for sf in r.struct_fields(["pos", "policy", "q_penalty", "q_no_penalty"]):
                                if sf == "pos":
                                    d["pos"] = [r.uint() for _ in r.struct_fields(["mask", "value"])]
                                elif sf == "policy":
                                    if r.length(4) != 7:
                                        raise ValueError("policy must have 7 entries")
                                    d["policy"] = [r.f32() for _ in range(7)]
                                else:
                                    d[sf] = r.f32()
                            samples.append(Sample(d["pos"][0], d["pos"][1], d["policy"], d["q_penalty"], d["q_no_penalty"]))
                results.append(GameResult(meta, samples))
        if r.i != len(cbor):
            raise ValueError("trailing bytes")
        return results
    except (IndexError, struct.error) as e:  # truncated input
        raise ValueError(f"invalid CBOR: {e}") from e


def _native():
    """(library, check): the host-side record functions of libc4a0_hip.so.  Missing library = ImportError with the build
    command (c4a0_amd._lib): there is no Python path behind it."""
    from . import _lib

    return _lib.lib(), _lib.check


def _sample_dtype():
    from .session import SAMPLE_DTYPE

    return SAMPLE_DTYPE


def _ids_table(reqs) -> np.ndarray:
    """uint64[n, 3] (game_id, player0_id, player1_id) from GameMetadata-like objects or an array of that shape."""
    if isinstance(reqs, np.ndarray):
        return np.ascontiguousarray(reqs, dtype=np.uint64).reshape(-1, 3)
    return np.array([(r.game_id, r.player0_id, r.player1_id) for r in reqs], dtype=np.uint64).reshape(-1, 3)


class PlayGamesResult:
    """pybridge.rs:58-158.  `score_policies` needs the external PascalPons solver and the
    rocksdb cache (rust/src/solver.rs) -- out of scope, raises NotImplementedError."""

    def __init__(self, results: Iterable[GameResult] = ()):  # pybridge.rs:67-70: empty constructor for unpickling
        self._results: List[GameResult] = list(results)
        self._lazy = None   # (ids uint64[n,3], records, counts uint32[n]): the arrays as handed over; objects are built on first access

    @classmethod
    def _from_records(cls, reqs, recs: np.ndarray, counts: np.ndarray) -> "PlayGamesResult":
        out = cls()
        ids = _ids_table(reqs)
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        if len(ids) != len(counts) or int(counts.sum(dtype=np.int64)) != len(recs):
            raise ValueError("records, counts and requests do not describe the same games")
        out._lazy = (ids, recs, counts)
        return out

    @property
    def results(self) -> List[GameResult]:
        """`Vec<GameResult>` (pybridge.rs:61-62).  A GPU run hands over ~500 k samples per second of
        play; the per-sample Python objects are only built when somebody asks for them."""
        if self._lazy is not None:
            ids, recs, counts = self._lazy
            self._lazy = None
            samples = Sample._bulk(recs)
            out, off = [], 0
            new = GameMetadata.__new__
            for (gid, p0, p1), n in zip(ids.tolist(), counts.tolist()):
                m = new(GameMetadata)
                m.game_id, m.player0_id, m.player1_id = gid, p0, p1
                out.append(GameResult(m, samples[off:off + n]))
                off += n
            self._results = out
        return self._results

    @results.setter
    def results(self, value):
        self._lazy = None
        self._results = list(value)

    def __len__(self) -> int:
        """Number of games (extension; does not build objects)."""
        return len(self._lazy[0]) if self._lazy is not None else len(self._results)

    def _tables(self):
        """(ids uint64[n,3], records, counts uint32[n]) of this result, whichever form it is held in."""
        if self._lazy is not None:
            return self._lazy
        counts = np.array([len(r.samples) for r in self._results], dtype=np.uint32)
        recs = np.zeros(int(counts.sum(dtype=np.int64)), dtype=_sample_dtype())
        ss = [s for r in self._results for s in r.samples]
        if ss:
            recs["game_id"] = np.repeat(np.array([r.metadata.game_id for r in self._results], dtype=np.uint64), counts.astype(np.int64))
            recs["mask"] = np.array([s.mask for s in ss], dtype=np.uint64)
            recs["value"] = np.array([s.value for s in ss], dtype=np.uint64)
            recs["policy"] = np.array([s.policy for s in ss], dtype=np.float32).reshape(-1, 7)
            recs["q_penalty"] = np.array([s.q_penalty for s in ss], dtype=np.float32)
            recs["q_no_penalty"] = np.array([s.q_no_penalty for s in ss], dtype=np.float32)
            ends = np.cumsum(counts.astype(np.int64))
            idx = np.arange(len(recs), dtype=np.int64) - np.repeat(ends - counts, counts.astype(np.int64))
            recs["meta"] = idx.astype(np.uint32)
            recs["meta"][ends[counts > 0] - 1] |= np.uint32(1 << 16)
        return _ids_table([r.metadata for r in self._results]), recs, counts

    def to_records(self):
        """(records, counts): the packed 64-byte sample records (c4a0_amd.session.SAMPLE_DTYPE) in result
        order and the number of samples of every game -- the form the GPU hands over (extension)."""
        _ids, recs, counts = self._tables()
        return recs, counts

    def to_arrays(self):
        """Bulk view for training code (extension; the reference only has per-sample `to_numpy`):
        (planes float32[N,2,6,7], policy float32[N,7], q_penalty float32[N], q_no_penalty float32[N],
        game_index int64[N]) over all samples in result order, without building Python objects."""
        _ids, recs, counts = self._tables()
        mask, value = recs["mask"], recs["value"]
        pol, qp, qn = recs["policy"].copy(), recs["q_penalty"].copy(), recs["q_no_penalty"].copy()
        gidx = np.repeat(np.arange(len(counts), dtype=np.int64), counts.astype(np.int64))
        bits = np.arange(42, dtype=np.uint64)[None, :]
        p0 = ((value[:, None] >> bits) & np.uint64(1)).astype(np.float32)
        p1 = (((mask & ~value)[:, None] >> bits) & np.uint64(1)).astype(np.float32)
        planes = np.concatenate([p0, p1], axis=1).reshape(-1, 2, N_ROWS, N_COLS)
        return planes, pol, qp, qn, gidx

    # -- serialisation (pybridge.rs:73-92): the library's host codec on the packed records (c4_records_to_cbor / c4_cbor_to_records)
    def to_cbor(self) -> bytes:
        import ctypes as C

        L, check = _native()
        ids, recs, counts = self._tables()
        recs = np.ascontiguousarray(recs)
        n = C.c_uint64()
        args = (ids.ctypes.data, counts.ctypes.data, len(counts), recs.ctypes.data, len(recs))
        check(L.c4_records_to_cbor(*args, None, 0, C.byref(n)))          # the document's size
        # the bytes object the caller gets, written in place (a bytes object may be filled by its creator before anyone else
        # sees it -- CPython's documented use of PyBytes_FromStringAndSize(NULL, n)): no staging buffer, no copy of 100 bytes per sample
        new_bytes = C.pythonapi.PyBytes_FromStringAndSize
        new_bytes.restype, new_bytes.argtypes = C.py_object, [C.c_char_p, C.c_ssize_t]
        as_ptr = C.pythonapi.PyBytes_AsString
        as_ptr.restype, as_ptr.argtypes = C.c_void_p, [C.py_object]
        out = new_bytes(None, n.value)
        check(L.c4_records_to_cbor(*args, as_ptr(out), n.value, C.byref(n)))
        return out

    @staticmethod
    def from_cbor(cbor: bytes) -> "PlayGamesResult":
        """pybridge.rs:80-92.  The result stays in record form (no per-sample objects) until `.results` is read."""
        import ctypes as C

        from ._lib import C4Error
        L, check = _native()
        src = np.frombuffer(cbor, dtype=np.uint8)          # no copy; bytes, bytearray, memoryview
        # one pass: a game takes at least 53 bytes of the document and a sample at least 59, which bounds the tables
        # (untouched pages of the over-sized arrays are never made resident)
        cap_games, cap_recs = len(src) // 53 + 1, len(src) // 59 + 1
        ids = np.empty((cap_games, 3), dtype=np.uint64)
        counts = np.empty(cap_games, dtype=np.uint32)
        recs = np.empty(cap_recs, dtype=_sample_dtype())
        n_games, n_recs = C.c_uint64(), C.c_uint64()
        try:
            check(L.c4_cbor_to_records(src.ctypes.data, len(src), ids.ctypes.data, counts.ctypes.data, cap_games,
                                       recs.ctypes.data, cap_recs, C.byref(n_games), C.byref(n_recs)))
        except C4Error as e:                               # the reference: ValueError via pyify_err (pybridge.rs:254-259)
            raise ValueError(str(e)) from None
        return PlayGamesResult._from_records(ids[: n_games.value], recs[: n_recs.value], counts[: n_games.value])

    def __getstate__(self) -> bytes:
        return self.to_cbor()

    def __setstate__(self, state: bytes) -> None:
        self._results = []
        self._lazy = PlayGamesResult.from_cbor(state)._lazy

    # -- pybridge.rs:95-106
    def __add__(self, other: "PlayGamesResult") -> "PlayGamesResult":
        if not isinstance(other, PlayGamesResult):
            raise TypeError("can only add PlayGamesResult")
        if self._lazy is not None and other._lazy is not None:      # stays in record form
            return PlayGamesResult._from_records(*(np.concatenate([a, b]) for a, b in zip(self._lazy, other._lazy)))
        return PlayGamesResult(self.results + other.results)

    # -- pybridge.rs:110-120: `results.shuffle(&mut StdRng::seed_from_u64(seed))`, whole games to one side.  The permutation
    # is rand's (c4_shuffle_games: seed_from_u64 -> ChaCha12 -> SliceRandom::shuffle), so the reference's training.py:207
    # gets the partition it would get from c4a0_rust.
    def split_train_test(self, train_frac: float, seed: int) -> Tuple[List[Sample], List[Sample]]:
        order = shuffled_game_order(len(self), seed)
        n_train = n_train_games(len(order), train_frac)
        if self._lazy is None:
            results = [self._results[i] for i in order.tolist()]
            return [s for r in results[:n_train] for s in r.samples], [s for r in results[n_train:] for s in r.samples]
        _ids, recs, counts = self._lazy                   # `self` stays in record form (and is not mutated: pybridge_test.py:22-39)
        idx, cut = split_record_indices(counts, train_frac, seed)
        samples = Sample._bulk(recs[idx])
        return samples[:cut], samples[cut:]

    def score_policies(self, solver_path: str, solver_book_path: str, solution_cache_path: str) -> float:
        raise NotImplementedError("score_policies needs the external c4solver binary and book (reference rust/src/solver.rs); out of scope")

    def unique_positions(self) -> int:  # pybridge.rs:150-157
        if self._lazy is not None:
            recs = self._lazy[1]
            return int(np.unique(np.stack([recs["mask"], recs["value"]], axis=1), axis=0).shape[0]) if len(recs) else 0
        return len({(s.mask, s.value) for r in self.results for s in r.samples})

    def __eq__(self, o):
        if not isinstance(o, PlayGamesResult):
            return False
        if self._lazy is None and o._lazy is None:
            return self._results == o._results
        (ia, ra, ca), (ib, rb, cb) = self._tables(), o._tables()
        if not (np.array_equal(ia, ib) and np.array_equal(ca, cb) and np.array_equal(ra["mask"], rb["mask"]) and np.array_equal(ra["value"], rb["value"])):
            return False
        return all(np.ascontiguousarray(ra[f]).tobytes() == np.ascontiguousarray(rb[f]).tobytes() for f in ("policy", "q_penalty", "q_no_penalty"))   # bit for bit, as Sample.__eq__


def n_train_games(n_games: int, train_frac: float) -> int:
    """pybridge.rs:113: `(results.len() as f32 * train_frac).round() as usize` -- the f32 product, Rust's f32::round (halves AWAY from
    zero: 5 games at 0.5 give 3; np.round is half-to-even) and the saturating `as usize` (NaN -> 0, negative -> 0, +inf / too large ->
    usize::MAX, then clamped to the number of games here, where the reference would panic on an out-of-range split)."""
    with np.errstate(over="ignore", invalid="ignore"):
        prod = float(np.float32(n_games) * np.float32(train_frac))
    if math.isnan(prod) or prod <= 0.0:
        return 0
    if math.isinf(prod):
        return n_games
    return min(n_games, int(math.floor(prod + 0.5)))


def split_record_indices(counts: np.ndarray, train_frac: float, seed: int) -> Tuple[np.ndarray, int]:
    """(idx, cut): records[idx[:cut]] are the training samples and records[idx[cut:]] the test samples of split_train_test, in its
    order (games in rand's shuffled order, whole games on one side, a game's samples in their own order)."""
    order = shuffled_game_order(len(counts), seed)
    n_train = n_train_games(len(order), train_frac)
    c64 = np.asarray(counts).astype(np.int64)
    starts = np.cumsum(c64) - c64
    oc = c64[order]
    idx = np.repeat(starts[order] - (np.cumsum(oc) - oc), oc) + np.arange(int(oc.sum()), dtype=np.int64)
    return idx, int(oc[:n_train].sum())


def shuffled_game_order(n_games: int, seed: int) -> np.ndarray:
    """order[i] = index of the game that `results.shuffle(&mut StdRng::seed_from_u64(seed))` (pybridge.rs:111-112) leaves at
    position i of a list of n_games (int64[n_games]); computed by the library's host function c4_shuffle_games."""
    import ctypes as C

    L, check = _native()
    order = np.empty(max(1, n_games), dtype=np.uint32)
    check(L.c4_shuffle_games(C.c_uint64(int(seed) & ((1 << 64) - 1)), n_games, order.ctypes.data))
    return order[:n_games].astype(np.int64)


def results_from_records(reqs, recs: np.ndarray, counts: np.ndarray) -> PlayGamesResult:
    """Wrap the packed sample records of `c4_session_drain_samples` (records of finished games in
    reqs order) and the per-game sample counts; `GameResult`/`Sample` objects are created lazily.
    `reqs`: GameMetadata-like objects, or a uint64[n, 3] array of (game_id, player0_id, player1_id)."""
    return PlayGamesResult._from_records(reqs, recs, counts)
