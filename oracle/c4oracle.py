"""ctypes binding of the CPU ORACLE (oracle/c4_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package (c4a0_amd/) never does.  See oracle/c4_oracle.h for the
parity-pinning statement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

# OpenMP workers must sleep, not spin, between the per-tick parallel regions: the evaluator
# callback runs on the main thread in between and would otherwise compete with spinning workers.
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "libc4oracle.so")


def _source_hash() -> str:
    import hashlib

    h = hashlib.sha256()
    for name in ("c4_oracle.c", "c4_oracle.h", "Makefile"):
        with open(os.path.join(_DIR, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:32]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (a second or two).  Returns the library path.  Stale = the hash of
    the sources (kept beside the library) differs: mtimes mean nothing after a snapshot copy."""
    stamp = _LIB_PATH + ".srchash"
    want = _source_hash()
    have = open(stamp).read().strip() if os.path.exists(stamp) and os.path.exists(_LIB_PATH) else None
    if force or have != want:
        subprocess.run(["make", "-C", _DIR, "-B", "libc4oracle.so"], check=True, capture_output=True)
        with open(stamp, "w") as f:
            f.write(want)
    return _LIB_PATH


class Pos(C.Structure):
    _fields_ = [("mask", C.c_uint64), ("value", C.c_uint64)]

    def key(self) -> Tuple[int, int]:
        return (int(self.mask), int(self.value))


class CSample(C.Structure):
    _fields_ = [("pos", Pos), ("policy", C.c_float * 7), ("q_penalty", C.c_float), ("q_no_penalty", C.c_float)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "sims", "sims_terminal_root", "select_levels", "select_levels_discarded", "backup_nodes", "expansions", "nodes_created", "moves")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class GameMetadataC(C.Structure):
    _fields_ = [("game_id", C.c_uint64), ("player0_id", C.c_uint64), ("player1_id", C.c_uint64)]


class SelfPlayStats(C.Structure):
    _fields_ = [("n_games", C.c_uint64), ("n_samples", C.c_uint64), ("nn_calls", C.c_uint64),
                ("nn_positions", C.c_uint64), ("tree", Counters)]


class EvalTableCtx(C.Structure):
    _fields_ = [("n", C.c_uint64), ("mask", C.c_void_p), ("value", C.c_void_p), ("out", C.c_void_p)]


EVAL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_float),
                      C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float))

NEXT_U32_FN = C.CFUNCTYPE(C.c_uint32, C.c_void_p)
P = C.POINTER

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    P = C.POINTER
    f32p, u32p = P(C.c_float), P(C.c_uint32)
    sig = {
        "c4o_make_move": (C.c_int, [P(Pos), C.c_int, P(Pos)]),
        "c4o_get": (C.c_int, [P(Pos), C.c_int, C.c_int]),
        "c4o_ply": (C.c_int, [P(Pos)]),
        "c4o_terminal_state": (C.c_int, [P(Pos)]),
        "c4o_terminal_value": (C.c_int, [P(Pos), C.c_float, f32p, f32p]),
        "c4o_legal_mask": (C.c_uint, [P(Pos)]),
        "c4o_flip_h": (None, [P(Pos), P(Pos)]),
        "c4o_write_planes": (None, [P(Pos), f32p]),
        "c4o_from_moves": (C.c_int, [P(C.c_int), C.c_int, P(Pos)]),
        "c4o_win_mask": (C.c_uint64, [C.c_int]),
        "c4o_pos_ops_batch": (None, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
        "c4o_random_positions": (None, [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
        "c4o_expf": (C.c_float, [C.c_float]),
        "c4o_logf": (C.c_float, [C.c_float]),
        "c4o_sweep_expf": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
        "c4o_sweep_logf": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
        "c4o_host_expf": (None, [f32p, f32p, C.c_size_t]),
        "c4o_host_logf": (None, [f32p, f32p, C.c_size_t]),
        "c4o_softmax7": (C.c_int, [f32p, f32p]),
        "c4o_apply_temperature": (None, [f32p, C.c_float, f32p]),
        "c4o_mask_policy": (None, [P(Pos), f32p]),
        "c4o_seed_from_u64": (None, [C.c_uint64, P(C.c_uint8)]),
        "c4o_chacha_block": (None, [P(C.c_uint8), C.c_uint64, C.c_int, u32p]),
        "c4o_rng_first_u32": (C.c_uint32, [C.c_uint64]),
        "c4o_weighted_index": (C.c_int, [f32p, C.c_uint32, P(C.c_int)]),
        "c4o_sample_move": (C.c_int, [C.c_uint64, C.c_int, f32p, C.c_float, P(C.c_int)]),
        "c4o_dirichlet": (None, [C.c_uint64, C.c_int, C.c_uint, C.c_float, f32p]),
        "c4o_partial_shuffle_with": (C.c_int, [C.c_uint64, C.c_uint64, NEXT_U32_FN, C.c_void_p, P(C.c_uint32)]),
        "c4o_shuffle_games": (C.c_int, [C.c_uint64, C.c_uint64, P(C.c_uint32)]),
        "c4o_self_play_set_dirichlet": (None, [C.c_float, C.c_float]),
        "c4o_game_set_dirichlet": (None, [C.c_void_p, C.c_float, C.c_float]),
        "c4o_game_new": (C.c_void_p, [P(Pos), C.c_uint64, C.c_uint64, C.c_uint64]),
        "c4o_game_free": (None, [C.c_void_p]),
        "c4o_game_root_pos": (None, [C.c_void_p, P(Pos)]),
        "c4o_game_leaf_pos": (None, [C.c_void_p, P(Pos)]),
        "c4o_game_leaf_model_id": (C.c_uint64, [C.c_void_p]),
        "c4o_game_on_received_policy": (C.c_int, [C.c_void_p, f32p, C.c_float, C.c_float, C.c_float, C.c_float]),
        "c4o_game_make_move": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
        "c4o_game_make_random_move": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
        "c4o_game_root_visit_count": (C.c_uint64, [C.c_void_p]),
        "c4o_game_root_policy": (None, [C.c_void_p, f32p]),
        "c4o_game_root_q_penalty": (C.c_float, [C.c_void_p]),
        "c4o_game_root_q_no_penalty": (C.c_float, [C.c_void_p]),
        "c4o_game_n_moves": (C.c_int, [C.c_void_p]),
        "c4o_game_error": (C.c_int, [C.c_void_p]),
        "c4o_game_counters": (None, [C.c_void_p, P(Counters)]),
        "c4o_game_to_result": (C.c_int, [C.c_void_p, C.c_float, P(CSample), C.c_int]),
        "c4o_game_step": (C.c_int, [C.c_void_p, f32p, C.c_float, C.c_float, C.c_uint64, C.c_float, C.c_float]),
        "c4o_self_play": (C.c_int, [P(GameMetadataC), C.c_uint64, C.c_int, C.c_uint64, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_int, P(CSample), P(C.c_uint64), P(SelfPlayStats)]),
        "c4o_self_play_async": (C.c_int, [P(GameMetadataC), C.c_uint64, C.c_int, C.c_uint64, C.c_float, C.c_float,
                                          C.c_void_p, C.c_void_p, C.c_int, P(CSample), P(C.c_uint64), P(SelfPlayStats)]),
        "c4o_hash_eval_pos": (None, [C.c_uint64, C.c_uint64, f32p, f32p, f32p]),
        "c4o_set_thread_pinning": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


# --------------------------------------------------------------------------- helpers
def _f7(v: Sequence[float]):
    return (C.c_float * 7)(*[float(x) for x in v])


def from_moves(cols: Sequence[int]) -> Pos:
    out = Pos()
    arr = (C.c_int * max(1, len(cols)))(*cols)
    if not lib().c4o_from_moves(arr, len(cols), C.byref(out)):
        raise ValueError("illegal move sequence")
    return out


def from_rows(rows_top_to_bottom: Sequence[str]) -> Pos:
    """Parse the reference's emoji board rendering (c4r.rs:395-431): first string is the
    TOP row; red = side to move ('value' bit), blue = opponent."""
    pos = Pos(0, 0)
    for row, line in enumerate(reversed(list(rows_top_to_bottom))):
        for col, ch in enumerate(line):
            bit = 1 << (row * 7 + col)
            if ch == "\U0001F534":  # red circle = Player
                pos.mask |= bit
                pos.value |= bit
            elif ch == "\U0001F535":  # blue circle = Opponent
                pos.mask |= bit
    return pos


def to_rows(pos: Pos) -> List[str]:
    """c4r.rs:395-413 Display."""
    out = []
    for row in range(5, -1, -1):
        s = ""
        for col in range(7):
            g = lib().c4o_get(C.byref(pos), row, col)
            s += "\U0001F534" if g == 1 else ("\U0001F535" if g == 0 else "⚫")
        out.append(s)
    return out


def make_move(pos: Pos, col: int) -> Optional[Pos]:
    out = Pos()
    return out if lib().c4o_make_move(C.byref(pos), col, C.byref(out)) else None


def terminal_state(pos: Pos) -> int:
    return lib().c4o_terminal_state(C.byref(pos))


def terminal_value(pos: Pos, c_ply_penalty: float):
    a, b = C.c_float(), C.c_float()
    t = lib().c4o_terminal_value(C.byref(pos), c_ply_penalty, C.byref(a), C.byref(b))
    return t, a.value, b.value


def legal_mask(pos: Pos) -> int:
    return lib().c4o_legal_mask(C.byref(pos))


def random_positions_np(n: int, seed: int = 1337):
    """n positions reachable by legal play (the reference's proptest strategy, c4r.rs:610-629) as (mask, value) uint64 arrays."""
    mask, value = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    lib().c4o_random_positions(n, seed, mask.ctypes.data, value.ctypes.data)
    return mask, value


def pos_ops_batch(mask: np.ndarray, value: np.ndarray, col: np.ndarray, c_ply_penalty: float):
    """legal mask, terminal state + values and make_move(col) of n positions in one C call ->
    (next_mask, next_value, legal int32[n], terminal int32[n], q float32[n, 2])."""
    n = len(mask)
    mask, value = np.ascontiguousarray(mask, dtype=np.uint64), np.ascontiguousarray(value, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    om, ov = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint64)
    ol, ot, oq = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32), np.zeros((n, 2), dtype=np.float32)
    lib().c4o_pos_ops_batch(mask.ctypes.data, value.ctypes.data, col.ctypes.data, n, c_ply_penalty, om.ctypes.data, ov.ctypes.data,
                            ol.ctypes.data, ot.ctypes.data, oq.ctypes.data)
    return om, ov, ol, ot, oq


def flip_h(pos: Pos) -> Pos:
    out = Pos()
    lib().c4o_flip_h(C.byref(pos), C.byref(out))
    return out


def planes(pos: Pos) -> np.ndarray:
    buf = np.zeros(84, dtype=np.float32)
    lib().c4o_write_planes(C.byref(pos), buf.ctypes.data_as(C.POINTER(C.c_float)))
    return buf.reshape(2, 6, 7)


def softmax7(logits: Sequence[float]) -> np.ndarray:
    out = (C.c_float * 7)()
    e = lib().c4o_softmax7(_f7(logits), out)
    if e:
        raise ValueError(f"softmax error {e}")
    return np.array(out[:], dtype=np.float32)


def apply_temperature(policy: Sequence[float], t: float) -> np.ndarray:
    out = (C.c_float * 7)()
    lib().c4o_apply_temperature(_f7(policy), t, out)
    return np.array(out[:], dtype=np.float32)


def partial_shuffle_with(items: Sequence[int], amount: int, next_u32) -> list:
    """rand's `partial_shuffle(amount)` of `items` (uint32 values) with `next_u32()` as the generator (pybridge.rs:110-116)."""
    arr = (C.c_uint32 * max(1, len(items)))(*items)
    cb = NEXT_U32_FN(lambda _ctx: next_u32() & 0xFFFFFFFF)
    if lib().c4o_partial_shuffle_with(len(items), amount, cb, None, arr):
        raise ValueError("slice too long")
    return list(arr[: len(items)])


def shuffle_games(seed: int, n_games: int) -> np.ndarray:
    """order[i] = index of the game `results.shuffle(&mut StdRng::seed_from_u64(seed))` leaves at position i."""
    out = np.zeros(max(1, n_games), dtype=np.uint32)
    if lib().c4o_shuffle_games(seed & ((1 << 64) - 1), n_games, out.ctypes.data_as(P(C.c_uint32))):
        raise ValueError("too many games")
    return out[:n_games]


def weighted_index(w: Sequence[float], u: int) -> int:
    idx = C.c_int()
    e = lib().c4o_weighted_index(_f7(w), u, C.byref(idx))
    if e:
        raise ValueError(f"weighted index error {e}")
    return idx.value


def sample_move(game_id: int, n_moves: int, policy: Sequence[float], temperature: float) -> int:
    col = C.c_int()
    e = lib().c4o_sample_move(game_id, n_moves, _f7(policy), temperature, C.byref(col))
    if e:
        raise ValueError(f"sample_move error {e}")
    return col.value


def seed_key(seed: int) -> bytes:
    key = (C.c_uint8 * 32)()
    lib().c4o_seed_from_u64(seed, key)
    return bytes(key)


def chacha_block(key: bytes, counter: int, rounds: int) -> List[int]:
    k = (C.c_uint8 * 32)(*key)
    out = (C.c_uint32 * 16)()
    lib().c4o_chacha_block(k, counter, rounds, out)
    return list(out)


def dirichlet(game_id: int, n_moves: int, legal: int, alpha: float) -> np.ndarray:
    out = (C.c_float * 7)()
    lib().c4o_dirichlet(game_id, n_moves, legal, alpha, out)
    return np.array(out[:], dtype=np.float32)


def hash_eval_pos(mask: int, value: int):
    lg = (C.c_float * 7)()
    a, b = C.c_float(), C.c_float()
    lib().c4o_hash_eval_pos(mask, value, lg, C.byref(a), C.byref(b))
    return np.array(lg[:], dtype=np.float32), a.value, b.value


class Game:
    """One MctsGame (mcts.rs:27-32)."""

    def __init__(self, pos: Optional[Pos] = None, game_id: int = 0, p0: int = 0, p1: int = 0):
        self._L = lib()
        p = pos if pos is not None else Pos(0, 0)
        self._g = self._L.c4o_game_new(C.byref(p), game_id, p0, p1)

    def __del__(self):
        if getattr(self, "_g", None):
            self._L.c4o_game_free(self._g)
            self._g = None

    def on_received_policy(self, logprobs, q_pen, q_nopen, c_exploration, c_ply_penalty) -> int:
        return self._L.c4o_game_on_received_policy(self._g, _f7(logprobs), q_pen, q_nopen, c_exploration, c_ply_penalty)

    def step(self, logprobs, q_pen, q_nopen, n_iter, c_exploration, c_ply_penalty) -> int:
        return self._L.c4o_game_step(self._g, _f7(logprobs), q_pen, q_nopen, n_iter, c_exploration, c_ply_penalty)

    def make_move(self, col, c_exploration) -> int:
        return self._L.c4o_game_make_move(self._g, col, c_exploration)

    def make_random_move(self, c_exploration, temperature) -> int:
        return self._L.c4o_game_make_random_move(self._g, c_exploration, temperature)

    def root_pos(self) -> Pos:
        p = Pos()
        self._L.c4o_game_root_pos(self._g, C.byref(p))
        return p

    def leaf_pos(self) -> Pos:
        p = Pos()
        self._L.c4o_game_leaf_pos(self._g, C.byref(p))
        return p

    def root_visit_count(self) -> int:
        return self._L.c4o_game_root_visit_count(self._g)

    def root_policy(self) -> np.ndarray:
        out = (C.c_float * 7)()
        self._L.c4o_game_root_policy(self._g, out)
        return np.array(out[:], dtype=np.float32)

    def root_q_penalty(self) -> float:
        return self._L.c4o_game_root_q_penalty(self._g)

    def root_q_no_penalty(self) -> float:
        return self._L.c4o_game_root_q_no_penalty(self._g)

    def n_moves(self) -> int:
        return self._L.c4o_game_n_moves(self._g)

    def error(self) -> int:
        return self._L.c4o_game_error(self._g)

    def counters(self) -> dict:
        c = Counters()
        self._L.c4o_game_counters(self._g, C.byref(c))
        return c.as_dict()

    def to_result(self, c_ply_penalty) -> List["SampleRec"]:
        buf = (CSample * 43)()
        n = self._L.c4o_game_to_result(self._g, c_ply_penalty, buf, 43)
        if n < 0:
            raise ValueError(f"to_result error {-n}")
        return [SampleRec.from_c(buf[i]) for i in range(n)]


@dataclass
class SampleRec:
    mask: int
    value: int
    policy: Tuple[float, ...]  # exact f32 values as Python floats
    q_penalty: float
    q_no_penalty: float

    @staticmethod
    def from_c(s: CSample) -> "SampleRec":
        return SampleRec(int(s.pos.mask), int(s.pos.value), tuple(float(x) for x in s.policy),
                         float(s.q_penalty), float(s.q_no_penalty))


def run_mcts(pos: Pos, n_iterations: int, c_exploration=4.0, c_ply_penalty=0.01,
             logprobs=None, q=(0.0, 0.0)):
    """mcts.rs:469-485 test helper `run_mcts` (constant evaluator)."""
    g = Game(pos)
    lp = _f7(logprobs if logprobs is not None else [np.float32(1.0) / np.float32(7.0)] * 7)
    L = lib()
    for _ in range(n_iterations):
        e = L.c4o_game_on_received_policy(g._g, lp, q[0], q[1], c_exploration, c_ply_penalty)
        if e:
            raise ValueError(f"mcts error {e}")
    return g.root_policy(), g.root_q_penalty(), g.root_q_no_penalty(), g


NpEval = Callable[[int, np.ndarray], Tuple[np.ndarray, np.ndarray, np.ndarray]]


def self_play(reqs: Sequence[Tuple[int, int, int]], max_nn_batch_size: int, n_mcts_iterations: int,
              c_exploration: float, c_ply_penalty: float, evaluator="uniform", n_threads: int = 1,
              dirichlet: Tuple[float, float] = (0.0, 0.0), topology: str = "lockstep"):
    """Oracle restatement of self_play.rs:39-129.

    `evaluator`: "uniform" | "zeros" | "hash" (built-in C evaluators) or a Python callable
    with the reference callback signature cb(model_id, float32[B,2,6,7]) ->
    (float32[B,7], float32[B], float32[B]) (pybridge.rs:170-198).
    Returns (dict game_id -> [SampleRec], stats dict).  Result order is per reqs order.

    topology="async" runs the reference's thread topology (c4o_self_play_async): this thread is the
    NN thread, n_threads - 1 worker threads run the MCTS jobs, evaluation and tree work overlap.
    Same samples either way.
    """
    L = lib()
    n = len(reqs)
    arr = (GameMetadataC * max(1, n))()
    for i, (gid, p0, p1) in enumerate(reqs):
        arr[i] = GameMetadataC(gid, p0, p1)
    out = (CSample * (43 * max(1, n)))()
    offs = (C.c_uint64 * (n + 1))()
    stats = SelfPlayStats()
    keep = None
    err: List[BaseException] = []
    ctx = None
    if isinstance(evaluator, str):
        fn = C.cast(getattr(L, {"uniform": "c4o_eval_uniform", "zeros": "c4o_eval_zeros", "hash": "c4o_eval_hash"}[evaluator]), C.c_void_p)
    elif isinstance(evaluator, tuple) and evaluator[0] == "table":
        # ("table", mask uint64[n], value uint64[n], out float32[n, 9]) sorted by (mask, value): c4o_eval_table, tier T3 at full size
        _tag, t_mask, t_value, t_out = evaluator
        t_mask, t_value = np.ascontiguousarray(t_mask, dtype=np.uint64), np.ascontiguousarray(t_value, dtype=np.uint64)
        t_out = np.ascontiguousarray(t_out, dtype=np.float32).reshape(-1, 9)
        assert len(t_mask) == len(t_value) == len(t_out)
        keep = (t_mask, t_value, t_out, EvalTableCtx(len(t_mask), t_mask.ctypes.data, t_value.ctypes.data, t_out.ctypes.data))
        ctx = C.cast(C.pointer(keep[3]), C.c_void_p)
        fn = C.cast(L.c4o_eval_table, C.c_void_p)
    else:
        def _cb(_ctx, model_id, nb, planes_p, lp_p, qp_p, qn_p):
            try:
                x = np.ctypeslib.as_array(planes_p, shape=(nb, 2, 6, 7)).copy()
                lp, qp, qn = evaluator(int(model_id), x)
                lp = np.ascontiguousarray(lp, dtype=np.float32).reshape(nb, 7)
                qp = np.ascontiguousarray(qp, dtype=np.float32).reshape(nb)
                qn = np.ascontiguousarray(qn, dtype=np.float32).reshape(nb)
                np.ctypeslib.as_array(lp_p, shape=(nb, 7))[:] = lp
                np.ctypeslib.as_array(qp_p, shape=(nb,))[:] = qp
                np.ctypeslib.as_array(qn_p, shape=(nb,))[:] = qn
                return 0
            except BaseException as e:  # noqa: BLE001 - surfaced below
                err.append(e)
                return 1

        keep = EVAL_FN(_cb)
        fn = C.cast(keep, C.c_void_p)
    L.c4o_self_play_set_dirichlet(float(dirichlet[0]), float(dirichlet[1]))  # extension; (0, 0) = off
    if topology not in ("lockstep", "async"):
        raise ValueError("topology must be 'lockstep' or 'async'")
    entry = L.c4o_self_play_async if topology == "async" else L.c4o_self_play
    rc = entry(arr, n, max_nn_batch_size, n_mcts_iterations, c_exploration, c_ply_penalty,
               fn, ctx, n_threads, out, offs, C.byref(stats))
    L.c4o_self_play_set_dirichlet(0.0, 0.0)
    if err:
        raise err[0]
    if rc:
        raise RuntimeError(f"oracle self_play error {rc}")
    res = {}
    for i, (gid, _p0, _p1) in enumerate(reqs):
        res.setdefault(gid, [])
        res[gid] = [SampleRec.from_c(out[j]) for j in range(offs[i], offs[i + 1])]
    st = {"n_games": int(stats.n_games), "n_samples": int(stats.n_samples), "nn_calls": int(stats.nn_calls),
          "nn_positions": int(stats.nn_positions), **stats.tree.as_dict()}
    return res, st
