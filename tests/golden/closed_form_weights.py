"""Weights from a closed-form integer formula, so that a fixture can hold inputs and outputs only.

`fill_closed_form(model)` overwrites every tensor of a ConnectFourNet `state_dict()` -- the reference's module
(tests/golden/make_nn_fixture.py, development container) and the build's restatement (the tests) alike, they have
the same keys and shapes -- with values that depend on nothing but the tensor's name and the element's flat index:

    u(i) = (mix64(i + (crc32(name) << 32)) >> 11) / 2^53        mix64 = the splitmix64 finaliser: three xor-shifts and
                                                                two multiplications modulo 2^64 (exact in float64)

    *.weight of a Conv2d / Linear   (u - 0.5) * 2 * sqrt(3 / fan_in)    variance 1 / fan_in
    *.bias of a Conv2d / Linear     (u - 0.5) * 0.2
    BatchNorm weight                0.5 + u          bias   (u - 0.5) * 0.4
    BatchNorm running_mean          (u - 0.5) * 0.6  running_var   0.5 + u      num_batches_tracked   untouched

No random number generator is involved: the same bits on every machine and torch version."""
import zlib

import numpy as np
import torch


def _u(name: str, n: int) -> np.ndarray:
    z = np.arange(n, dtype=np.uint64) + (np.uint64(zlib.crc32(name.encode())) << np.uint64(32))
    z = z + np.uint64(0x9E3779B97F4A7C15)                      # unsigned 64-bit arithmetic wraps modulo 2^64
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def fill_closed_form(model: torch.nn.Module) -> None:
    bn_prefixes = {n for n, m in model.named_modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)}
    sd = model.state_dict()
    for name, t in sd.items():
        if name.endswith("num_batches_tracked"):
            continue
        u = _u(name, t.numel())
        prefix, leaf = name.rsplit(".", 1)
        if prefix in bn_prefixes:
            v = {"weight": 0.5 + u, "bias": (u - 0.5) * 0.4, "running_mean": (u - 0.5) * 0.6, "running_var": 0.5 + u}[leaf]
        elif leaf == "weight":
            fan_in = t.numel() // t.shape[0]
            v = (u - 0.5) * 2.0 * np.sqrt(3.0 / fan_in)
        else:
            v = (u - 0.5) * 0.2
        t.copy_(torch.from_numpy(v.astype(np.float32)).reshape(t.shape))
