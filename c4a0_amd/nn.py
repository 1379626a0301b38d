"""ConnectFourNet inference for the self-play evaluator (reference src/c4a0/nn.py).

Only the forward pass is in scope: the reference evaluates leaves through
`ConnectFourNet.forward_numpy` (nn.py:119-130), a host round trip per batch.  Here the same
network runs as PyTorch-ROCm on the device that holds the trees, on the tensors the HIP step
kernel reads and writes.

* `ConnectFourNet` has the reference's module tree, so `state_dict()` keys are identical
  (`conv.0.*`, `conv.{i}.block.{0,1,2}.*`, `fc_policy.*`, `fc_value.*`; nn.py:64-100,184-195)
  and a reference checkpoint loads unchanged.
* `InferenceNet` is the evaluator: BatchNorm folded into the preceding conv/linear (eval-mode
  statistics), weights in bf16 (or f32), log-softmax/tanh in f32, outputs written into
  caller-provided tensors; optionally captured in a HIP graph.
"""
from __future__ import annotations

import threading
import warnings
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

N_COLS, N_ROWS = 7, 6


@dataclass
class ModelConfig:  # nn.py:16-38 (training-only fields omitted)
    n_residual_blocks: int = 1
    conv_filter_size: int = 32
    n_policy_layers: int = 4
    n_value_layers: int = 2


class ResidualBlock(nn.Module):  # nn.py:184-195: x + ReLU(BN(Conv(Conv(x))))
    def __init__(self, n_channels: int):
        super().__init__()
        self.block = nn.Sequential(
            nn.Conv2d(n_channels, n_channels, kernel_size=3, padding=1),
            nn.Conv2d(n_channels, n_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(n_channels),
            nn.ReLU(),
        )

    def forward(self, x):
        return x + self.block(x)


class ConnectFourNet(nn.Module):
    def __init__(self, config: ModelConfig):
        super().__init__()
        self.config = config
        c = config.conv_filter_size
        self.conv = nn.Sequential(  # nn.py:64-70: first conv has no BN / ReLU
            nn.Conv2d(2, c, kernel_size=3, padding=1),
            *[ResidualBlock(c) for _ in range(config.n_residual_blocks)],
        )
        fc = c * N_ROWS * N_COLS  # nn.py:72,132-138
        self.fc_policy = nn.Sequential(  # nn.py:75-86
            *[nn.Sequential(nn.Linear(fc, fc), nn.BatchNorm1d(fc), nn.ReLU()) for _ in range(config.n_policy_layers - 1)],
            nn.Linear(fc, N_COLS),
            nn.LogSoftmax(dim=1),
        )
        self.fc_value = nn.Sequential(  # nn.py:89-100
            *[nn.Sequential(nn.Linear(fc, fc), nn.BatchNorm1d(fc), nn.ReLU()) for _ in range(config.n_value_layers - 1)],
            nn.Linear(fc, 2),
            nn.Tanh(),
        )

    def forward(self, x) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:  # nn.py:109-117
        x = self.conv(x)
        x = x.reshape(x.shape[0], -1)  # "b c h w -> b (c h w)"
        policy_logprobs = self.fc_policy(x)
        q = self.fc_value(x)
        return policy_logprobs, q[:, 0], q[:, 1]


def _fold_bn(weight: torch.Tensor, bias: torch.Tensor, bn: nn.modules.batchnorm._BatchNorm):
    """Eval-mode BN(y) = (y - mean) / sqrt(var + eps) * gamma + beta folded into y = W x + b."""
    scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
    shape = [-1] + [1] * (weight.dim() - 1)
    w = weight.detach().double() * scale.reshape(shape)
    b = (bias.detach().double() - bn.running_mean.detach().double()) * scale + bn.bias.detach().double()
    return w.float(), b.float()


def pack_tower_weights(conv_w, conv_b, channels: int):
    """Re-order the (BN-folded) conv weights into the MFMA A-fragment order the HIP tower kernel
    reads (c4a0_amd/csrc/c4_conv_tower.hip): for v_mfma_f32_16x16x32_bf16 lane l holds
    A[row = l & 15][k = 8 (l >> 4) + j], j = 0..7.

      conv_w[0]  [C, 2, 3, 3]  -> w0 [3 steps][C/16][64 lanes][8]: k-step s, k-group g <-> tap 4 s + g,
                                  element j <-> input channel j (only j < 2 non-zero)
      conv_w[1:] [C, C, 3, 3]  -> w  [layer][9 taps][C/16][C/32][64 lanes][8]:
                                  W[co = 16 m + (l & 15)][ci = 32 kc + 8 (l >> 4) + j][tap]
      conv_b     [C] each      -> bias [1 + 2 n_blocks][C] float32
    """
    c = channels
    mt, kc = c // 16, c // 32
    w0 = conv_w[0].float().reshape(c, 2, 9)                       # [co, ci, tap]
    p0 = torch.zeros(3, mt, 4, 16, 8)                             # [s, m, g, co_l, j]
    for s in range(3):
        for g in range(4):
            tap = 4 * s + g
            if tap < 9:
                p0[s, :, g, :, 0:2] = w0[:, :, tap].reshape(mt, 16, 2)
    p0 = p0.reshape(3, mt, 64, 8)
    layers = []
    for w in conv_w[1:]:
        wt = w.float().reshape(c, c, 9).permute(2, 0, 1)          # [tap, co, ci]
        wt = wt.reshape(9, mt, 16, kc, 4, 8).permute(0, 1, 3, 4, 2, 5)  # [tap, m, kc, g, co_l, j]
        layers.append(wt.reshape(9, mt, kc, 64, 8))
    pw = torch.stack(layers) if layers else torch.zeros(0, 9, mt, kc, 64, 8)
    bias = torch.stack([b.float() for b in conv_b])
    return p0.contiguous(), pw.contiguous(), bias.contiguous()


class EvaluatorFallbackWarning(UserWarning):
    """An InferenceNet that leaves the hand-written HIP kernels (channels not 32 / 64, or not bf16, or not on a HIP device)."""


class InferenceNet:
    """Device-resident evaluator: planes[G,2,6,7] -> (policy_logprobs[G,7] f32, q[G,2] f32).

    `hip_tower=True` (default on a HIP device in bf16 with 32 or 64 channels) runs the conv tower
    as the hand-written MFMA kernel `c4_conv_tower_bf16`; otherwise PyTorch-ROCm convs + the library GEMM -- a path
    whose low bits depend on the batch (`batch_invariant` False), slower, and never the one the parity suite certifies.
    Leaving the hand-written kernels is therefore LOUD: `strict=True` refuses (ValueError), the default warns once per
    construction (EvaluatorFallbackWarning) and `path` says which one runs ("hip" | "torch"; the bench line carries it)."""

    latency_mode = False  # True while ONE session plays alone on the device (api._play sets it): the narrow layers of a
                          # > 1 024-row batch then use the 128 x 96 tile (12.7 vs 16.8 us alone at 1 700 rows; beside a
                          # second session's kernels the fat 128 x 192 tile wins, c4_head_gemm.hip) and the 32-channel
                          # tower 8 boards per workgroup up to 2 048 boards (27.3 -> 19.1 us alone at 2 048)
    graph_safe = True  # forward() is pure device work on caller-owned outputs: may be captured in a HIP graph
    stage_hook = None  # optional callable(stage): 0 = before the tower is launched, 2 = after it, 1 = after the first hidden layer's
                       # GEMM is launched, 3 + i = after the policy head's i-th further layer (session.capture_pair records /
                       # waits cross-stream events there)

    def __init__(self, model: ConnectFourNet, device: torch.device, dtype: torch.dtype = torch.bfloat16,
                 hip_tower: Optional[bool] = None, gemm: Optional[str] = None, gemm_config=None, tower_config: int = 0,
                 strict: bool = False):
        """gemm: "hip" (the hand-written MFMA GEMM, default with the HIP tower) or "hipblaslt" (PyTorch's library GEMM, an
        A/B switch: its low bits depend on the batch shape).  gemm_config: c4_linear_bf16's tile configuration, one number or
        "wide,narrow" (the merged 2F-wide first layer, the F-wide layers); tower_config: c4_conv_tower_bf16's workgroup
        shape; 0 / None = automatic.  Measurement switches are constructor arguments: nothing is read from the environment."""
        self.device = torch.device(device)
        self.dtype = dtype
        model = model.eval()
        self.channels = model.config.conv_filter_size
        can_tower = self.device.type == "cuda" and dtype == torch.bfloat16 and self.channels in (32, 64)
        if hip_tower and not can_tower:
            raise ValueError("hip_tower needs a HIP device, bf16 and 32 or 64 channels")
        self.hip_tower = can_tower if hip_tower is None else bool(hip_tower)
        if not self.hip_tower and hip_tower is None and self.device.type == "cuda":
            # nobody asked for PyTorch's kernels (hip_tower=False does: the A/B switch of the tests): say so
            why = (f"{self.channels} channels (the HIP tower and GEMM are written for 32 and 64)" if dtype == torch.bfloat16
                   else f"dtype {dtype} (the HIP kernels compute in bf16)")
            msg = (f"InferenceNet: {why}: this evaluator runs on PyTorch's convolutions and the library GEMM, not on the "
                   "hand-written HIP kernels -- slower, and a position's low bits depend on the batch it is evaluated in")
            if strict:
                raise ValueError(msg + " (strict=True)")
            warnings.warn(msg, EvaluatorFallbackWarning, stacklevel=2)
        conv0 = model.conv[0]
        self.conv_w = [conv0.weight.detach().float()]
        self.conv_b = [conv0.bias.detach().float()]
        self.n_blocks = len(model.conv) - 1
        for blk in list(model.conv)[1:]:
            c1, c2, bn = blk.block[0], blk.block[1], blk.block[2]
            self.conv_w.append(c1.weight.detach().float())
            self.conv_b.append(c1.bias.detach().float())
            w, b = _fold_bn(c2.weight, c2.bias, bn)
            self.conv_w.append(w)
            self.conv_b.append(b)

        def head(seq):
            ws, bs = [], []
            mods = list(seq)
            for m in mods[:-2]:
                w, b = _fold_bn(m[0].weight, m[0].bias, m[1])
                ws.append(w)
                bs.append(b)
            ws.append(mods[-2].weight.detach().float())
            bs.append(mods[-2].bias.detach().float())
            return ws, bs

        self.pol_w, self.pol_b = head(model.fc_policy)
        self.val_w, self.val_b = head(model.fc_value)
        if self.hip_tower:
            from . import _lib

            self._L = _lib.lib()
            w0, w, bias = pack_tower_weights(self.conv_w, self.conv_b, self.channels)
            self.tw0 = w0.to(self.device, torch.bfloat16).contiguous()
            self.tw = w.to(self.device, torch.bfloat16).contiguous()
            self.tbias = bias.to(self.device, torch.float32).contiguous()
            # the tower emits [cell][channel]; the reference flattens "c h w" (nn.py:111): permute
            # the input dimension of each head's first Linear once instead of the activations
            c = self.channels
            perm = lambda wt: wt.reshape(wt.shape[0], c, 42).permute(0, 2, 1).reshape(wt.shape[0], 42 * c)
            self.pol_w[0] = perm(self.pol_w[0])
            self.val_w[0] = perm(self.val_w[0])
        self.pol_b32 = self.pol_b[-1].to(self.device, torch.float32).contiguous()
        self.val_b32 = self.val_b[-1].to(self.device, torch.float32).contiguous()
        self.fused_epilogue = self.device.type == "cuda" and hasattr(torch, "_addmm_activation")
        # Hidden layers of the heads: "hip" = the hand-written MFMA GEMM (c4_linear_bf16), whose result for
        # a position does not depend on the batch or the row it sits in -- the default wherever the HIP
        # tower runs; "hipblaslt" = PyTorch's library GEMM (kept for A/B timing: its low bits depend on
        # the batch shape).
        gemm = gemm or ("hip" if self.hip_tower else "hipblaslt")
        if gemm not in ("hip", "hipblaslt"):
            raise ValueError("gemm must be 'hip' or 'hipblaslt'")
        if gemm == "hip" and not (self.hip_tower and (42 * self.channels) % 192 == 0):
            raise ValueError("gemm='hip' needs the HIP tower (bf16, 32 or 64 channels on a HIP device)")
        self.gemm = gemm
        if strict and gemm != "hip":
            raise ValueError("InferenceNet(strict=True): gemm='hipblaslt' leaves the hand-written GEMM (its low bits depend on the batch)")
        self._np_lock = threading.Lock()   # forward_numpy keeps ONE pinned slot, stream and graph set per net
        # every kernel of the evaluator computes a row from that row alone in one fixed order: a position's outputs do
        # not depend on the batch (session.narrow_if_worthwhile may then narrow whenever it likes)
        self.batch_invariant = self.hip_tower and gemm == "hip"
        # tile configuration: one number, or "wide,narrow" (the merged 2F-wide first layer, the F-wide layers); 0 = automatic
        cfg = str(gemm_config if gemm_config is not None else "0").split(",")
        self.gemm_config = (int(cfg[0]), int(cfg[-1]))
        self.tower_config = int(tower_config)   # c4_conv_tower_bf16's config, 0 = automatic
        mv = lambda ts: [t.to(self.device, dtype).contiguous() for t in ts]
        self.conv_w = [w.to(self.device, dtype).contiguous(memory_format=torch.channels_last) for w in self.conv_w]
        self.conv_b = mv(self.conv_b)
        self.pol_w, self.pol_b, self.val_w, self.val_b = mv(self.pol_w), mv(self.pol_b), mv(self.val_w), mv(self.val_b)
        self.merged_w1 = self.merged_b1 = None
        if len(self.pol_w) > 1 and len(self.val_w) > 1:
            self.merged_w1 = torch.cat([self.pol_w[0], self.val_w[0]], dim=0).contiguous()
            self.merged_b1 = torch.cat([self.pol_b[0], self.val_b[0]], dim=0).contiguous()
        self._bias32 = {}   # f32 copies of the hidden layers' biases for the HIP GEMM's epilogue, by bias tensor
        if self.gemm == "hip":
            for b in self.pol_b[:-1] + self.val_b[:-1] + ([self.merged_b1] if self.merged_b1 is not None else []):
                self._bias32[b.data_ptr()] = b.float().contiguous()

    @property
    def path(self) -> str:
        """Which kernels run: "hip" = the hand-written tower, GEMM and output kernels; "torch" = any PyTorch / library kernel in the chain."""
        return "hip" if (self.hip_tower and self.gemm == "hip") else "torch"

    @torch.no_grad()
    def tower(self, planes: torch.Tensor, latency: Optional[bool] = None) -> torch.Tensor:
        """Conv tower -> flattened features [G, 42*C] (cell-major when hip_tower, else "c h w").  latency: this call's
        answer to `latency_mode` (None = the attribute)."""
        latency = self.latency_mode if latency is None else latency
        if self.hip_tower:
            from ._lib import check
            import ctypes as C

            x = planes if planes.dtype == torch.bfloat16 else planes.to(torch.bfloat16)
            x = x.contiguous()
            g = x.shape[0]
            out = torch.empty((g, 42 * self.channels), dtype=torch.bfloat16, device=self.device)
            check(self._L.c4_conv_tower_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(self.tw0.data_ptr()),
                                             C.c_void_p(self.tw.data_ptr()), C.c_void_p(self.tbias.data_ptr()),
                                             g, self.channels, self.n_blocks, C.c_void_p(out.data_ptr()),
                                             # alone on the device (latency_mode): 8 boards per workgroup from 1 025 to 2 048 boards
                                             # (up to 1 024 boards the automatic choice is already the narrow-launch shape: 2 or 4 boards per workgroup)
                                             self.tower_config or (2 if (latency and self.channels == 32 and 1024 < g <= 2048) else 0),
                                             C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            return out
        x = planes.to(self.dtype)
        x = F.conv2d(x, self.conv_w[0], self.conv_b[0], padding=1)
        for i in range(self.n_blocks):
            y = F.conv2d(x, self.conv_w[1 + 2 * i], self.conv_b[1 + 2 * i], padding=1)
            y = F.conv2d(y, self.conv_w[2 + 2 * i], self.conv_b[2 + 2 * i], padding=1)
            x = x + F.relu(y)
        return x.reshape(x.shape[0], -1)

    @torch.no_grad()
    def forward_hidden(self, planes: torch.Tensor, latency: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Everything but the heads' output layers: planes -> (policy head's last hidden activations, value head's), bf16
        [G, F] row views (column stride 1).  The output layers follow in `forward`, or inside the session's fused output +
        step launch (DeviceSession.round, c4_session_step_head_out)."""
        hook = self.stage_hook
        if hook is not None:
            hook(0)
        x = self.tower(planes, latency)
        if hook is not None:
            hook(2)                # after the tower is launched (capture_pair's offset_stage measurements)
        if self.merged_w1 is not None:
            # first hidden layer of BOTH heads as one GEMM (same input, N = 2F): better tile occupancy.  (Round 4 tried the two
            # halves as two launches, the value head's on a side stream beside the policy head's second layer, so that a
            # session's chain holds an F-wide layer instead of the 2F-wide one: same bits, but every cross-stream edge
            # costs more than the 3.5 us it saves -- the bench halved, 28.0 -> 13.7 k games/s -- and ROCm 7.2's
            # hipStreamEndCapture crashes on a fork from a non-origin stream, tools/capture_nested_fork_repro.py.)
            h = self._linear_relu(x, self.merged_w1, self.merged_b1, latency=latency)
            if hook is not None:
                hook(1)
            f = self.merged_w1.shape[0] // 2
            p, v = h[:, :f], h[:, f:]
            pol_rest, val_rest = list(zip(self.pol_w[1:-1], self.pol_b[1:-1])), list(zip(self.val_w[1:-1], self.val_b[1:-1]))
        else:
            p = v = x
            pol_rest, val_rest = list(zip(self.pol_w[:-1], self.pol_b[:-1])), list(zip(self.val_w[:-1], self.val_b[:-1]))
            if hook is not None:   # no merged first layer (a head without hidden layers): the heavy half ends with the tower
                hook(1)
        for i, (w, b) in enumerate(pol_rest):
            p = self._linear_relu(p, w, b, latency=latency)
            if hook is not None:
                hook(3 + i)      # after each narrow policy layer (capture_pair's offset_stage)
        for w, b in val_rest:
            v = self._linear_relu(v, w, b, latency=latency)
        return p, v

    @property
    def fused_step_ok(self) -> bool:
        """The session may run this evaluator's output layers inside its step launch (c4_session_step_head_out), i.e. call
        forward_hidden() + head_out_operands() INSTEAD of forward().  Only while forward() is InferenceNet's own: a subclass
        that overrides `forward` / `__call__` (counting calls, post-processing logits, observers) must see every evaluation,
        so for it the session keeps evaluate() -> step() (the same answer as setting
        DeviceSession.fuse_output_step = False)."""
        own = type(self).forward is InferenceNet.forward and type(self).__call__ is InferenceNet.__call__
        return own and bool(self.hip_tower) and (42 * self.channels) % 1344 == 0

    def head_out_operands(self):
        """(w_policy, w_value, b_policy f32, b_value f32) of the output layers, as the HIP output kernels take them."""
        return self.pol_w[-1], self.val_w[-1], self.pol_b32, self.val_b32

    @torch.no_grad()
    def forward(self, planes: torch.Tensor, out_logprobs: Optional[torch.Tensor] = None,
                out_q: Optional[torch.Tensor] = None, latency: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        p, v = self.forward_hidden(planes, latency)
        if self.hip_tower:
            # both output layers + log-softmax + tanh in one HIP launch, written in place
            from ._lib import check
            import ctypes as C

            g = p.shape[0]
            lp = out_logprobs if out_logprobs is not None else torch.empty((g, 7), dtype=torch.float32, device=self.device)
            q = out_q if out_q is not None else torch.empty((g, 2), dtype=torch.float32, device=self.device)
            assert p.stride(1) == 1 and v.stride(1) == 1
            check(self._L.c4_head_out_bf16(C.c_void_p(p.data_ptr()), C.c_void_p(v.data_ptr()),
                                           C.c_void_p(self.pol_w[-1].data_ptr()), C.c_void_p(self.val_w[-1].data_ptr()),
                                           C.c_void_p(self.pol_b32.data_ptr()), C.c_void_p(self.val_b32.data_ptr()),
                                           g, p.shape[1], p.stride(0), v.stride(0), C.c_void_p(lp.data_ptr()), C.c_void_p(q.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            return lp, q
        p = F.linear(p, self.pol_w[-1], self.pol_b[-1]).float()
        v = F.linear(v, self.val_w[-1], self.val_b[-1]).float()
        if out_logprobs is None:
            lp = torch.log_softmax(p, dim=1)
        else:
            lp = torch.log_softmax(p, dim=1, out=out_logprobs)
        q = torch.tanh(v) if out_q is None else torch.tanh(v, out=out_q)
        return lp, q

    # ---------------------------------------------------------------- the reference's numpy entry point
    _NP_BUCKET = 128

    @torch.no_grad()
    def forward_numpy(self, x):
        """`ConnectFourNet.forward_numpy` (reference src/c4a0/nn.py:119-130): float32 numpy positions [B, 2, 6, 7] ->
        (policy_logprobs float32 [B, 7], q_penalty float32 [B], q_no_penalty float32 [B]), fresh C-contiguous arrays --
        what the reference's callers hand to `play_games` as `lambda model_id, x: model.forward_numpy(x)`
        (training.py:179-189), so a caller that swaps its model for this class keeps its callback.

        The reference's version is a host round trip around ~35 eager launches.  Here ONE HIP-graph replay and one stream
        synchronisation: the graph (captured once per batch size rounded up to a multiple of 128 rows: the kernels compute a
        row from that row alone, so padding rows change nothing) is c4_planes_from_f32 + tower + GEMMs + output kernel, and
        the output kernel writes into pinned host memory itself.  A batch that already lives in pinned host memory -- what
        `play_games` hands its callback -- is read by the first kernel over PCIe where it is (its address travels in a
        pinned word, c4_f32_batch); any other array takes one copy into a device buffer first."""
        import ctypes as C
        import numpy as np

        from ._lib import check

        x = np.ascontiguousarray(x, dtype=np.float32)
        b = int(x.shape[0])
        if b == 0:
            return np.zeros((0, 7), np.float32), np.zeros((0,), np.float32), np.zeros((0,), np.float32)
        if not (self.hip_tower and self.device.type == "cuda"):
            lp, q = self.forward(torch.from_numpy(x).to(self.device))
            lp, q = lp.float().cpu().numpy(), q.float().cpu().numpy()
            return np.ascontiguousarray(lp), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])
        if x.shape[1:] != (2, 6, 7):
            raise ValueError(f"forward_numpy: positions must be [B, 2, 6, 7], got {x.shape}")
        bucket = -(-b // self._NP_BUCKET) * self._NP_BUCKET
        # ONE pinned slot, stream and graph set per net: two threads calling forward_numpy on the same net (a tournament's
        # callbacks, the cpu_baseline's evaluator thread beside the main thread) take turns -- the reference's
        # forward_numpy is re-entrant, this one is serialised
        with self._np_lock:
            return self._forward_numpy_locked(x, b, bucket)

    def _forward_numpy_locked(self, x, b: int, bucket: int):
        import ctypes as C
        import numpy as np

        from ._lib import check

        st = getattr(self, "_np", None)
        if st is None or st["cap"] < bucket:
            cap = max(2048, 1 << (bucket - 1).bit_length())
            st = self._np = {"cap": cap, "graphs": {}, "stream": torch.cuda.Stream(device=self.device),
                             "in": torch.zeros((cap, 2, 6, 7), dtype=torch.float32, device=self.device),
                             "planes": torch.zeros((cap, 2, 6, 7), dtype=torch.bfloat16, device=self.device),
                             "slot": torch.zeros(2, dtype=torch.int64).pin_memory(),      # c4_f32_batch {data, n_boards}
                             "h_lp": torch.zeros((cap, 7), dtype=torch.float32).pin_memory(),
                             "h_q": torch.zeros((cap, 2), dtype=torch.float32).pin_memory()}
            st["slot_np"] = st["slot"].numpy()
        stream = st["stream"]
        xt = torch.from_numpy(x) if x.flags.writeable else None
        direct = xt is not None and x.ctypes.data % 16 == 0 and xt.is_pinned()
        with torch.cuda.stream(stream):
            g = st["graphs"].get(bucket)
            if g is None:
                from .session import CAPTURE_ERROR_MODE

                def body():
                    check(self._L.c4_planes_from_f32(C.c_void_p(st["slot"].data_ptr()), None, 0, C.c_void_p(st["planes"].data_ptr()), bucket,
                                                     C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
                    # the output kernel's stores go to pinned host memory: no copy after it.  latency=True: a host round
                    # trip, this forward has the chip to itself (an argument of the call, not a mutation of the shared net)
                    self.forward(st["planes"][:bucket], out_logprobs=st["h_lp"][:bucket], out_q=st["h_q"][:bucket], latency=True)

                st["slot_np"][0], st["slot_np"][1] = st["in"].data_ptr(), 0     # warm-up and capture run on empty boards
                for _ in range(2):      # warm-up outside the capture (lazy module loads, LDS opt-ins)
                    body()
                stream.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=stream, capture_error_mode=CAPTURE_ERROR_MODE):
                    body()
                st["graphs"][bucket] = g
            if direct:
                st["slot_np"][0] = x.ctypes.data
            else:
                st["in"][:b].copy_(torch.from_numpy(x), non_blocking=True)
                st["slot_np"][0] = st["in"].data_ptr()
            st["slot_np"][1] = b
            g.replay()
        stream.synchronize()
        lp, q = st["h_lp"].numpy()[:b], st["h_q"].numpy()[:b]
        return lp.copy(), np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])

    def _alone_config(self, m: int, n: int, k: int, latency: Optional[bool] = None) -> int:
        """Tile configuration of a hidden layer when ONE session has the device to itself (latency_mode).  Above 1 024 rows the
        automatic choice is the fat tile that wins beside a second session's kernels; alone, up to 1 728 rows the 96 x 96 tile (at
        most 2 x 256 workgroups, two per CU) and beyond that the 128 x 96 tile for the F-wide layers are faster.  Up to 1 024 rows the
        wave-specialised forms of the small tiles (two or four wavefronts that only issue the DMA pieces) are 3-15 % faster ALONE and
        5 % slower beside a second session (profiles/r04_gemm_configs.txt (3), profiles/r05_small_rows_ab.txt), so they too are asked
        for here and nowhere else.  Every configuration computes the same bits.  0 = the library's automatic choice."""
        if not (self.latency_mode if latency is None else latency):
            return 0
        wide = n > k
        if k >= 2048:   # the 64-channel net: the automatic 256 x 192 tile for the 2F-wide layer, 128 x 192 for the F-wide ones
            return 0 if (m <= 1024 or wide) else 11   # (alone at 2 048 rows: 31 us against 49)
        if m <= 1024:
            if not self.use_loader_waves:
                return 0
            # (thresholds = where a tile's grid stops being one wave of workgroups; re-probed in round 5 with every configuration,
            # tools/gemm_sweep.py, profiles/r05_gemm_configs.txt (6))
            if wide:
                return 41 if m <= 384 else (42 if m <= 576 else (44 if m <= 864 else 43))
            return 41 if m <= 512 else 42
        if m <= 1728:
            if wide and self.use_loader_waves and self.wide_tiles_r5:
                # the 2F-wide layer (round 5, profiles/r05_gemm_configs.txt (5)): 128 x 96 while its 9 x 28 tiles are one wave of workgroups
                # (10.7-11.1 us against 14-15), then 192 x 96 with two loading wavefronts (<= 9 x 28 tiles up to 1 728 rows, a quarter
                # fewer operand bytes per CU than 96 x 96: 13.5-15.2 us against 15-17.4)
                return 43 if m <= 1152 else 59
            return 44 if self.use_loader_waves else 23   # 96 x 96, four computing (+ four loading) wavefronts
        # alone, the wave-specialised forms: 128 x 192 on 8 + 4 wavefronts for the 2F-wide layer, 128 x 96 on 4 + 2 for the F-wide ones
        if self.use_loader_waves:
            return 35 if wide else 43
        return 11 if wide else 10

    use_loader_waves = True   # False: no wave-specialised (loader-wavefront) forms anywhere (A/B)
    wide_tiles_r5 = True      # False: round 4's choice for the 2F-wide layer between 1 025 and 1 728 rows (A/B)

    def _pick_config(self, m: int, n: int, k: int, latency: Optional[bool] = None) -> int:
        return self._alone_config(m, n, k, latency)

    def _linear_relu(self, x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None,
                     latency: Optional[bool] = None) -> torch.Tensor:
        """ReLU(x W^T + b): the hand-written MFMA GEMM, or (gemm="hipblaslt") the library's with the bias
        and ReLU in its epilogue where available."""
        if self.gemm == "hip":
            from ._lib import check
            import ctypes as C

            assert x.stride(1) == 1 and w.is_contiguous()
            m, n, k = x.shape[0], w.shape[0], w.shape[1]
            y = out if out is not None else torch.empty((m, n), dtype=torch.bfloat16, device=self.device)
            check(self._L.c4_linear_bf16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(self._bias32[b.data_ptr()].data_ptr()),
                                         C.c_void_p(y.data_ptr()), m, n, k, x.stride(0), y.stride(0), 1,
                                         self.gemm_config[0 if n > k else 1] or self._pick_config(m, n, k, latency),
                                         C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
            return y
        if self.fused_epilogue:
            return torch._addmm_activation(b, x, w.t(), use_gelu=False)
        return F.relu(F.linear(x, w, b))

    __call__ = forward


class GraphedEvaluator:
    """The evaluator captured once in a HIP graph: one replay per step instead of ~25 eager
    launches.  Reads `planes`, writes `logprobs` and `q` (the session's bound tensors)."""

    def __init__(self, net: InferenceNet, planes: torch.Tensor, logprobs: torch.Tensor, q: torch.Tensor, warmup: int = 3):
        self.net, self.planes, self.logprobs, self.q = net, planes, logprobs, q
        side = torch.cuda.Stream(device=planes.device)
        side.wait_stream(torch.cuda.current_stream(planes.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                net.forward(planes, out_logprobs=logprobs, out_q=q)
        torch.cuda.current_stream(planes.device).wait_stream(side)
        torch.cuda.synchronize(planes.device)
        self.graph = torch.cuda.CUDAGraph()
        from .session import CAPTURE_ERROR_MODE
        with torch.cuda.graph(self.graph, capture_error_mode=CAPTURE_ERROR_MODE):
            net.forward(planes, out_logprobs=logprobs, out_q=q)

    def __call__(self, _planes: torch.Tensor):
        self.graph.replay()
        return self.logprobs, self.q


def flops_per_leaf(cfg: ModelConfig) -> int:
    """2*MAC count of one forward."""
    c, f = cfg.conv_filter_size, cfg.conv_filter_size * 42
    conv = 2 * 42 * 9 * 2 * c + cfg.n_residual_blocks * 2 * (2 * 42 * 9 * c * c)
    pol = (cfg.n_policy_layers - 1) * 2 * f * f + 2 * 7 * f
    val = (cfg.n_value_layers - 1) * 2 * f * f + 2 * 2 * f
    return conv + pol + val
