"""The C ABI driven from C: tests/abi_consumer.c is compiled with gcc against include/c4a0_hip.h only
(no Python, no torch in that process), plays whole games with a constant "network" and prints its
samples; every one must equal the oracle's (evaluator kind "zeros").  Proves the struct layouts,
ownership rules and call order of INTEGRATION.md from the side a Rust `extern "C"` binding comes from
(reference rust/src/pybridge.rs:20-53, rust/src/lib.rs:23-41)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_consumer(out_dir) -> str:
    exe = os.path.join(str(out_dir), "abi_consumer")
    pkg = os.path.join(ROOT, "c4a0_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "abi_consumer.c"), "-o", exe, "-L" + pkg, "-l:libc4a0_hip.so", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + pkg + ",-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_consumer_compiles_as_plain_c(tmp_path):
    """CPU: the header + consumer are valid C11 and link against the library (no GPU call)."""
    from c4a0_amd.csrc import build as hip_build
    hip_build.build()
    exe = build_consumer(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("n_games,n_slots,n_iter", [(20, 8, 20), (5, 16, 7)])
def test_c_consumer_equals_the_oracle(tmp_path, n_games, n_slots, n_iter):
    from oracle import c4oracle as O

    exe = build_consumer(tmp_path)
    r = subprocess.run([exe, str(n_games), str(n_slots), str(n_iter)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    head = dict(zip(lines[0].split()[0::2], map(int, lines[0].split()[1::2])))
    got = {}
    for l in lines[1:]:
        f = l.split()
        gid, idx, flags, mask, value = int(f[0]), int(f[1]), int(f[2]), int(f[3], 16), int(f[4], 16)
        pol = b"".join(struct.pack("<I", int(x, 16)) for x in f[5:12])
        got.setdefault(gid, []).append((idx, flags, mask, value, pol, struct.pack("<I", int(f[12], 16)), struct.pack("<I", int(f[13], 16))))
    reqs = [(500 + 3 * i, 0, 0) for i in range(n_games)]
    want, st = O.self_play(reqs, 64, n_iter, 6.6, 0.01, "zeros")
    assert head["games"] == n_games and head["samples"] == sum(len(v) for v in want.values()) == len(lines) - 1
    assert head["expansions"] == st["expansions"]
    for gid, _, _ in reqs:
        mine = got[gid]
        assert [m[0] for m in mine] == list(range(len(mine)))                      # records in index order
        assert [m[1] for m in mine] == [0] * (len(mine) - 1) + [1]                  # the last one is the terminal sample
        ora = [(s.mask, s.value, np.array(s.policy, dtype=np.float32).tobytes(), np.float32(s.q_penalty).tobytes(),
                np.float32(s.q_no_penalty).tobytes()) for s in want[gid]]
        assert [m[2:] for m in mine] == ora, f"game {gid}"
