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


def build_consumer(out_dir, name: str = "abi_consumer") -> str:
    exe = os.path.join(str(out_dir), name)
    pkg = os.path.join(ROOT, "c4a0_amd")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", name + ".c"), "-o", exe, "-L" + pkg, "-l:libc4a0_hip.so", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + pkg + ",-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_consumer_compiles_as_plain_c(tmp_path):
    """CPU: the header + consumer are valid C11 and link against the library (no GPU call)."""
    from c4a0_amd.csrc import build as hip_build
    hip_build.build()
    for name in ("abi_consumer", "abi_consumer_nn"):
        exe = build_consumer(tmp_path, name)
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


def _parse(stdout: str):
    lines = stdout.strip().splitlines()
    head = dict(zip(lines[0].split()[0::2], map(int, lines[0].split()[1::2])))
    got = {}
    for l in lines[1:]:
        f = l.split()
        pol = b"".join(struct.pack("<I", int(x, 16)) for x in f[5:12])
        got.setdefault(int(f[0]), []).append((int(f[3], 16), int(f[4], 16), pol, struct.pack("<I", int(f[12], 16)), struct.pack("<I", int(f[13], 16))))
    return head, got


@pytest.mark.gpu
@pytest.mark.parametrize("blocks,channels,pol,val", [(1, 32, 4, 2), (2, 64, 2, 3)])
def test_c_host_runs_the_network_itself(tmp_path, blocks, channels, pol, val):
    """tests/abi_consumer_nn.c: sessions AND the bf16 ResNet evaluator driven from plain C through the C ABI
    (c4_conv_tower_bf16, c4_linear_bf16, c4_head_out_bf16, c4_session_*), weights from a file -- no Python, no torch
    in that process.  Its samples are byte-identical to play_games(evaluator=InferenceNet) in this one: the evaluator
    is a function of the position, whoever launches its kernels."""
    import torch

    import c4a0_amd
    from c4a0_amd.nn import ConnectFourNet, InferenceNet, ModelConfig

    torch.manual_seed(77)
    net = InferenceNet(ConnectFourNet(ModelConfig(blocks, channels, pol, val)), torch.device("cuda:0"), dtype=torch.bfloat16)
    assert net.hip_tower and net.gemm == "hip" and net.merged_w1 is not None
    F = 42 * channels
    blobs = [net.tw0, net.tw, net.tbias, net.merged_w1, net._bias32[net.merged_b1.data_ptr()]]
    for ws, bs in ((net.pol_w, net.pol_b), (net.val_w, net.val_b)):
        for w, b in zip(ws[1:-1], bs[1:-1]):
            blobs += [w, net._bias32[b.data_ptr()]]
    blobs += [net.pol_w[-1], net.val_w[-1], net.pol_b32, net.val_b32]
    path = os.path.join(str(tmp_path), "weights.bin")
    with open(path, "wb") as f:
        f.write(struct.pack("<6I", channels, blocks, F, len(net.pol_w) - 2, len(net.val_w) - 2, 0))
        for t in blobs:
            raw = t.contiguous().cpu().view(torch.uint8).numpy().tobytes()
            f.write(struct.pack("<Q", len(raw)))
            f.write(raw)
    n_games, n_slots, n_iter = 24, 16, 12
    exe = build_consumer(tmp_path, "abi_consumer_nn")
    r = subprocess.run([exe, path, str(n_games), str(n_slots), str(n_iter)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    head, got = _parse(r.stdout)
    reqs = [c4a0_amd.GameMetadata(900 + i, 0, 0) for i in range(n_games)]
    stats = {}
    res = c4a0_amd.play_games(reqs, 64, n_iter, 6.6, 0.01, evaluator=net, resident_games=n_slots, stats=stats)
    want = {r_.metadata.game_id: [(x.mask, x.value, x.policy.tobytes(), x.q_penalty.tobytes(), x.q_no_penalty.tobytes()) for x in r_.samples]
            for r_ in res.results}
    assert head["games"] == n_games and head["samples"] == stats["samples"] and head["expansions"] == stats["expansions"]
    assert got == want
    # ... and the same job as ONE call from C: c4_play_games_bf16, the library's own host loop (round 6)
    r = subprocess.run([exe, path, str(n_games), str(n_slots), str(n_iter), "native"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    head2, got2 = _parse(r.stdout)
    assert head2 == head and got2 == want
