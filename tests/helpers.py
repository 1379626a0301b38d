"""Shared helpers of the parity tests (test infrastructure)."""
from __future__ import annotations

import numpy as np

P1, P2, P3, P4, MOD = 1000003, 998244353, 19260817, 1000000007, 2147483647


def planes_to_pos_np(planes: np.ndarray):
    """float[B,2,6,7] -> (mask, value) uint64 arrays (inverse of c4r.rs:378-392)."""
    b = planes.reshape(planes.shape[0], 2, 42) != 0
    w = (np.uint64(1) << np.arange(42, dtype=np.uint64))
    value = (b[:, 0, :].astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)
    opp = (b[:, 1, :].astype(np.uint64) * w).sum(axis=1, dtype=np.uint64)
    return value | opp, value


def hash_eval_np(_model_id, planes: np.ndarray):
    """numpy twin of oracle c4o_hash_eval_pos (reference callback signature)."""
    mask, value = planes_to_pos_np(planes)
    mask = mask.astype(np.int64)
    value = value.astype(np.int64)
    h = ((value & 0x1FFFFF) * P1 + (value >> 21) * P2 + (mask & 0x1FFFFF) * P3 + (mask >> 21) * P4) % MOD
    c = np.arange(7, dtype=np.int64)[None, :]
    hc = (h[:, None] * (2 * c + 3) + 7919 * c) % 1000003
    logits = ((hc & 63) - 32).astype(np.float32) / np.float32(8.0)
    qp = (((h >> 5) & 255) - 128).astype(np.float32) / np.float32(128.0)
    qn = (((h >> 13) & 255) - 128).astype(np.float32) / np.float32(128.0)
    return logits, qp, qn


def hash_eval_torch(planes):
    """torch twin (device evaluator): planes[G,2,6,7] -> (logits[G,7] f32, q[G,2] f32)."""
    import torch

    g = planes.shape[0]
    b = (planes.reshape(g, 2, 42) != 0).to(torch.int64)
    w = (torch.ones(42, dtype=torch.int64, device=planes.device) << torch.arange(42, dtype=torch.int64, device=planes.device))
    value = (b[:, 0, :] * w).sum(dim=1)
    mask = value | (b[:, 1, :] * w).sum(dim=1)
    h = ((value & 0x1FFFFF) * P1 + (value >> 21) * P2 + (mask & 0x1FFFFF) * P3 + (mask >> 21) * P4) % MOD
    c = torch.arange(7, dtype=torch.int64, device=planes.device)[None, :]
    hc = (h[:, None] * (2 * c + 3) + 7919 * c) % 1000003
    logits = ((hc & 63) - 32).to(torch.float32) / 8.0
    qp = (((h >> 5) & 255) - 128).to(torch.float32) / 128.0
    qn = (((h >> 13) & 255) - 128).to(torch.float32) / 128.0
    return logits, torch.stack([qp, qn], dim=1)


class GraphSafeHashEval:
    """hash_eval_torch as a graph-safe device evaluator (pure device work written into the caller's
    tensors): lets the parity tests drive the HIP-graph / concurrent-session paths of play_games with
    an evaluator whose answers are exact integers of the position (independent of batch shape)."""
    graph_safe = True
    dtype = None

    def __call__(self, planes, out_logprobs=None, out_q=None):
        lp, q = hash_eval_torch(planes)
        if out_logprobs is None:
            return lp, q
        out_logprobs.copy_(lp)
        out_q.copy_(q)
        return out_logprobs, out_q


def uniform_eval_torch(planes):
    """self_play.rs:391-403 UniformEvalPos on device."""
    import torch

    g = planes.shape[0]
    lp = torch.full((g, 7), float(np.float32(1.0) / np.float32(7.0)), dtype=torch.float32, device=planes.device)
    return lp, torch.zeros((g, 2), dtype=torch.float32, device=planes.device)


def samples_by_game(recs: np.ndarray):
    """structured sample array -> {game_id: [(mask, value, policy bytes, q_pen bits, q_nopen bits), ...]} in index order."""
    out = {}
    order = np.lexsort((recs["meta"] & 0xFFFF, recs["game_id"]))
    for r in recs[order]:
        out.setdefault(int(r["game_id"]), []).append(
            (int(r["mask"]), int(r["value"]), r["policy"].astype(np.float32).tobytes(),
             np.float32(r["q_penalty"]).tobytes(), np.float32(r["q_no_penalty"]).tobytes()))
    return out


def oracle_samples_by_game(res: dict):
    out = {}
    for gid, samples in res.items():
        out[int(gid)] = [(s.mask, s.value, np.array(s.policy, dtype=np.float32).tobytes(),
                          np.float32(s.q_penalty).tobytes(), np.float32(s.q_no_penalty).tobytes()) for s in samples]
    return out


def random_positions(n: int, seed: int = 1337):
    """The reference's `random_pos` strategy (c4r.rs:610-629), vectorised in numpy: play up to
    `k` random columns from the empty board, skipping illegal ones, stopping at terminal."""
    from oracle import c4oracle as O
    import random

    rng = random.Random(seed)
    out = []
    L = O.lib()
    import ctypes as C
    while len(out) < n:
        pos = O.Pos(0, 0)
        for _ in range(rng.randrange(0, 60)):
            if L.c4o_terminal_state(C.byref(pos)) != 0:
                break
            mov = rng.randrange(7)
            if (L.c4o_legal_mask(C.byref(pos)) >> mov) & 1:
                nx = O.Pos()
                L.c4o_make_move(C.byref(pos), mov, C.byref(nx))
                pos = nx
            out.append((int(pos.mask), int(pos.value)))  # every prefix is a reachable position too
            if len(out) >= n:
                break
    return out[:n]


EVIDENCE = []


def evidence(line: str) -> None:
    """A line for the end of the pytest run (tests/conftest.py pytest_terminal_summary): how much a parity test compared, so that
    the count lands in the driver's record of the run and not only in a builder-side log."""
    EVIDENCE.append(line)
