"""Build libc4a0_hip.so (the C-ABI library) in-tree with hipcc for gfx950.

    python c4a0_amd/csrc/build.py [--force]

-ffp-contract=off: the tree arithmetic must round exactly like the reference's separate f32
operations (rust never contracts a*b+c); the one fused operation the algorithm needs is written
as __builtin_fma.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libc4a0_hip.so")
SRCS = [os.path.join(HERE, "c4_session.hip"), os.path.join(HERE, "c4_conv_tower.hip")]
DEPS = SRCS + [os.path.join(HERE, "c4_device.hpp"), os.path.join(os.path.dirname(PKG), "include", "c4a0_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True builds libc4a0_hip_diag.so with in-kernel phase stamps (tools/phase_profile.py)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = OUT.replace(".so", "_diag.so") if diag else OUT
    stale = force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in DEPS)
    if stale:
        cmd = [hipcc] + FLAGS + (["-DC4_PHASE_STAMPS"] if diag else []) + SRCS + ["-o", out]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
