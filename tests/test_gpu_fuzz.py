"""Randomised HIP-vs-oracle parity under the driver's clock: a time-boxed run of tools/fuzz_parity.py with a
FIXED seed -- random job shapes and options (slots, games, n, exploration constants, Dirichlet noise, evaluation cache, HIP-graph or
eager, one simulation per step), a quarter of the jobs through the numpy callback with one to four models, a fifth with the bf16
network (fused graph path against the eager stand-alone kernels).  Every sample of every game must equal the oracle's bit for bit,
and jobs that panic in the reference (tiny n, mcts.rs:196-200) must fail with the same status on both sides.  The multi-hour soaks of
earlier rounds live in profiles/r0*_fuzz_parity.txt; this one is the part the round-end GPU suite re-runs."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed,seconds,env", [
    (20260005, 28, {}),                                                            # the mix: tree / numpy callback / bf16 network jobs
    (20260006, 22, {"FUZZ_TREE_ONLY": "1", "FUZZ_RECLAIM_SHARE": "1", "FUZZ_DIRICHLET_SHARE": "1"}),   # every job: reclaimed arenas + Dirichlet noise
])
def test_time_boxed_fuzz_parity_fixed_seed(seed, seconds, env):
    from tests.helpers import evidence
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), str(seconds), str(seed)], cwd=ROOT, capture_output=True,
                       text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    m = re.search(r"fuzz parity ok: (\d+) jobs \((\d+) of them through the numpy callback.*?; (\d+) with the bf16 network.*?; (\d+) through the native host loop.*?; (\d+) on arenas reclaimed.*?; (\d+) with Dirichlet noise\), (\d+) games", r.stdout)
    assert m, r.stdout[-1500:]
    jobs, cb_jobs, net_jobs, native_jobs, reclaimed_jobs, dirichlet_jobs, games = (int(g) for g in m.groups())
    evidence(f"fuzz seed {seed}: " + r.stdout.strip().splitlines()[-1])
    # an MI355X runs ~7 jobs per second of the mix (profiles/r04_fuzz_parity.txt: 4 132 jobs in 10 minutes); far fewer means the soak did not really run
    if env:
        assert jobs >= 40 and reclaimed_jobs >= 0.5 * jobs and dirichlet_jobs == jobs and games >= 300, r.stdout[-500:]
    else:
        assert jobs >= 40 and cb_jobs >= 5 and net_jobs >= 3 and native_jobs >= 3 and reclaimed_jobs >= 6 and games >= 400, r.stdout[-500:]
