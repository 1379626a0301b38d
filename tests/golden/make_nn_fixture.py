"""Generates tests/golden/nn_fixture.npz and tests/golden/nn_fixture_1x32.npz by importing the REFERENCE's own
src/c4a0/nn.py.

Runs only in the development container (needs /root/reference); the committed .npz is the
fixture.  The reference module needs pytorch_lightning / torchmetrics / loguru / c4a0_rust,
which are absent here, so four stub modules provide exactly the names nn.py touches
(LightningModule = torch.nn.Module + save_hyperparameters/log; dummy metrics; the three
board constants).  Nothing from the reference is copied: the fixture is a state_dict, inputs
and the outputs the reference's forward produced.

nn_fixture_1x32.npz (round 4) is at a width the HIP kernels accept -- 1 block x 32 channels, 2 policy / 2 value
layers -- so that the hand-written tower, GEMM and output kernel are compared with outputs of the reference itself
(tests/test_gpu_nn.py).  Its 3.6 M weights come from a closed-form integer formula (closed_form_weights.py, shared
with the test), so the file holds inputs and outputs only.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_fixture.npz")
OUT_1X32 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn_fixture_1x32.npz")


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


def random_positions(n, seed, max_plies=30):
    """positions from random legal play, encoded exactly as c4r.rs:378-392"""
    rng = np.random.default_rng(seed)
    xs = []
    for _ in range(n):
        mask = value = 0
        for _ in range(int(rng.integers(0, max_plies))):
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
    return np.stack(xs)


def fixture_1x32(ConnectFourNet, ModelConfig):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from closed_form_weights import fill_closed_form

    cfg = ModelConfig(n_residual_blocks=1, conv_filter_size=32, n_policy_layers=2, n_value_layers=2,
                      lr_schedule={0: 1e-3}, l2_reg=0.0)
    model = ConnectFourNet(cfg)
    with torch.no_grad():
        fill_closed_form(model)
    x = random_positions(96, seed=23, max_plies=42)
    lp, qp, qn = model.forward_numpy(x)  # reference nn.py:119-130
    np.savez_compressed(OUT_1X32, x=x, policy_logprobs=lp, q_penalty=qp, q_no_penalty=qn,
                        cfg=np.array([1, 32, 2, 2], dtype=np.int64))
    print("wrote", OUT_1X32, os.path.getsize(OUT_1X32), "bytes; weights: closed_form_weights.fill_closed_form")


def main():
    stub_modules()
    sys.path.insert(0, REF)
    from c4a0.nn import ConnectFourNet, ModelConfig  # the reference's own module

    fixture_1x32(ConnectFourNet, ModelConfig)

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
