"""Device-resident training tensors out of packed sample records.

The reference's training loop turns every `Sample` into numpy arrays on the host
(`Sample.to_numpy`, types.rs:125-147; `SampleDataModule`, training.py) and ships them to the GPU
again.  Here the finished games' 64-byte records (`DeviceSession.pack_samples_device`, the buffer
the RCCL all-gather moves) become the four training tensors without leaving HBM; `flip_h`
(types.rs:115-122, c4r.rs:289-299), the reference's mirror augmentation, is a flip of the column
axis.  Plain torch ops on device tensors: this is glue, not a hot kernel.
"""
from __future__ import annotations

from typing import Tuple

import torch


def records_to_tensors(records: torch.Tensor, flip_h: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """records uint8[N, 64] (c4_sample_rec, include/c4a0_hip.h) ->
    (pos float32[N,2,6,7], policy float32[N,7], q_penalty float32[N], q_no_penalty float32[N]),
    the tuple `Sample.to_numpy()` yields per sample, on the records' device."""
    if records.dtype != torch.uint8 or records.dim() != 2 or records.shape[1] != 64:
        raise ValueError("records must be uint8[N, 64]")
    r = records.contiguous()
    w64 = r.view(torch.int64)            # [N, 8]: game_id, mask, value, ...
    f32 = r.view(torch.float32)          # [N, 16]: ..., policy at 6..12, q_penalty 13, q_no_penalty 14, meta 15
    mask, value = w64[:, 1], w64[:, 2]
    bits = torch.arange(42, dtype=torch.int64, device=r.device)
    p0 = ((value[:, None] >> bits) & 1).to(torch.float32)                # side to move (c4r.rs:378-392)
    p1 = (((mask & ~value)[:, None] >> bits) & 1).to(torch.float32)
    pos = torch.stack([p0, p1], dim=1).reshape(-1, 2, 6, 7)
    policy = f32[:, 6:13].clone()
    if flip_h:
        pos = pos.flip(3)
        policy = policy.flip(1)
    return pos.contiguous(), policy.contiguous(), f32[:, 13].clone(), f32[:, 14].clone()


def training_tensors(session, augment_flip_h: bool = True):
    """All samples of a session's finished games as device tensors, followed (like the reference's
    `SampleDataModule`, which adds `s.flip_h()` for every sample) by their mirror images."""
    recs = session.pack_samples_device()
    out = records_to_tensors(recs)
    if not augment_flip_h:
        return out
    mirrored = records_to_tensors(recs, flip_h=True)
    return tuple(torch.cat([a, b], dim=0) for a, b in zip(out, mirrored))


def split_train_test_tensors(result, train_frac: float, seed: int, device=None, augment_flip_h: bool = True):
    """`games.split_train_test(train_frac, seed)` followed by `SampleDataModule(train, test, ...)` (reference
    src/c4a0/training.py:207 and 317-333) without a Python object per sample: the SAME partition (whole games to one side, rand's
    shuffle on `seed`, `PlayGamesResult.split_train_test`) and the same order of samples, as two tuples of tensors on `device` --
    (pos float32[N,2,6,7], policy float32[N,7], q_penalty float32[N], q_no_penalty float32[N]) for training and for validation --
    each followed, like `SampleDataModule`, by the mirror images of all its samples (`Sample.flip_h`)."""
    import numpy as np

    from .results import split_record_indices

    _ids, recs, counts = result._tables()
    idx, cut = split_record_indices(counts, train_frac, seed)
    rows = torch.from_numpy(np.ascontiguousarray(recs[idx]).view(np.uint8).reshape(-1, 64))
    if device is not None:
        rows = rows.to(device)
    out = []
    for part in (rows[:cut], rows[cut:]):
        t = records_to_tensors(part)
        if augment_flip_h:
            m = records_to_tensors(part, flip_h=True)
            t = tuple(torch.cat([a, b], dim=0) for a, b in zip(t, m))
        out.append(t)
    return out[0], out[1]
