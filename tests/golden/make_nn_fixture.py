"""Generates tests/golden/nn_fixture.npz by importing the REFERENCE's own src/c4a0/nn.py.

Runs only in the development container (needs /root/reference); the committed .npz is the
fixture.  The reference module needs pytorch_lightning / torchmetrics / loguru / c4a0_rust,
which are absent here, so four stub modules provide exactly the names nn.py touches
(LightningModule = torch.nn.Module + save_hyperparameters/log; dummy metrics; the three
board constants).  Nothing from the reference is copied: the fixture is a state_dict, inputs
and the outputs the reference's forward produced.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_fixture.npz")


def stub_modules():
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

        @property
        def device(self):
            return next(self.parameters()).device

    pl.LightningModule = LightningModule
    tm = types.ModuleType("torchmetrics")

    class _Metric(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tm.KLDivergence = _Metric
    tm.MeanSquaredError = _Metric
    lg = types.ModuleType("loguru")
    lg.logger = types.SimpleNamespace(info=lambda *a, **k: None)
    cr = types.ModuleType("c4a0_rust")
    cr.N_COLS, cr.N_ROWS, cr.BUF_N_CHANNELS = 7, 6, 2
    sys.modules.update({"pytorch_lightning": pl, "torchmetrics": tm, "loguru": lg, "c4a0_rust": cr})


def main():
    stub_modules()
    sys.path.insert(0, REF)
    from c4a0.nn import ConnectFourNet, ModelConfig  # the reference's own module

    torch.manual_seed(1337)
    cfg = ModelConfig(n_residual_blocks=2, conv_filter_size=4, n_policy_layers=3, n_value_layers=2,
                      lr_schedule={0: 1e-3}, l2_reg=0.0)
    model = ConnectFourNet(cfg)
    # non-trivial BatchNorm statistics so that eval-mode BN (and its folding) is exercised
    g = torch.Generator().manual_seed(7)
    for m in model.modules():
        if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.3)
            m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)
    # inputs: positions from random legal play, encoded exactly as c4r.rs:378-392
    rng = np.random.default_rng(11)
    xs = []
    for _ in range(24):
        mask = value = 0
        for _ in range(int(rng.integers(0, 30))):
            col = int(rng.integers(0, 7))
            h = bin(mask & (0x810204081 << col)).count("1")
            if h == 6:
                continue
            bit = 1 << (7 * h + col)
            mask |= bit
            value = ~(value | bit) & mask
        pl0 = [(value >> i) & 1 for i in range(42)]
        pl1 = [((mask & ~value) >> i) & 1 for i in range(42)]
        xs.append(np.array(pl0 + pl1, dtype=np.float32).reshape(2, 6, 7))
    x = np.stack(xs)
    lp, qp, qn = model.forward_numpy(x)  # reference nn.py:119-130
    sd = {"sd/" + k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(OUT, x=x, policy_logprobs=lp, q_penalty=qp, q_no_penalty=qn,
                        cfg=np.array([2, 4, 3, 2], dtype=np.int64), **sd)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(sd), "state_dict entries")


if __name__ == "__main__":
    main()
