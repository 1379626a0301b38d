import os
import random
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: exhaustive sweeps, minutes of CPU (run explicitly with -m slow)")


@pytest.fixture(autouse=True)
def _seed():
    # reference tests/c4a0_tests/conftest.py:8-11 seeds python/numpy/torch with 1337
    random.seed(1337)
    np.random.seed(1337)
    try:
        import torch

        torch.manual_seed(1337)
    except Exception:  # pragma: no cover
        pass


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """What the parity tests of this run compared (tests.helpers.evidence): printed after the results, also with -q."""
    try:
        from tests.helpers import EVIDENCE
    except Exception:  # pragma: no cover
        return
    if EVIDENCE:
        terminalreporter.write_line("parity evidence of this run:")
        for line in EVIDENCE:
            terminalreporter.write_line("  " + line)
