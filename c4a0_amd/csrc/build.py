"""Build libc4a0_hip.so (the C-ABI library) in-tree with hipcc for gfx950.

    python c4a0_amd/csrc/build.py [--force]

-ffp-contract=off: the tree arithmetic must round exactly like the reference's separate f32
operations (rust never contracts a*b+c); the one fused operation the algorithm needs is written
as __builtin_fma.

Staleness is decided by CONTENT, not by mtime (a snapshot copy resets mtimes): the sha256 of the
sources, headers and flags is compiled into the library (`c4_source_hash()`, and the marker string
`c4a0-src-hash:<hex>` in its bytes); `build()` recompiles whenever that differs from the sources
in the tree, and `c4a0_amd._lib` refuses a library whose hash does not match the sources beside it.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.path.join(PKG, "libc4a0_hip.so")
SRCS = [os.path.join(HERE, "c4_session.hip"), os.path.join(HERE, "c4_conv_tower.hip"), os.path.join(HERE, "c4_head_gemm.hip"), os.path.join(HERE, "c4_results_host.hip"), os.path.join(HERE, "c4_selfplay_host.hip")]
DEPS = SRCS + [os.path.join(HERE, "c4_device.hpp"), os.path.join(HERE, "c4_host.hpp"), os.path.join(HERE, "c4_head_out.hpp"), os.path.join(HERE, "c4_timeline.hpp"),
               os.path.join(os.path.dirname(PKG), "include", "c4a0_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function",
         # scalar kernel arguments (up to 16 dwords) arrive in SGPRs at wavefront launch instead of by a scalar load from
         # the kernarg segment: the hot kernels take their pointers and sizes as leading scalar arguments for this
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]
MARKER = b"c4a0-src-hash:"


def source_hash(extra_flags=()) -> str:
    """sha256 over the flags and the bytes of every file the library is built from."""
    h = hashlib.sha256(" ".join(FLAGS + list(extra_flags)).encode())
    for d in DEPS:
        h.update(os.path.basename(d).encode() + b"\0")
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:32]


def embedded_hash(path: str):
    """The hash compiled into a built library, or None (missing file / library from before the marker)."""
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    i = blob.find(MARKER)
    if i < 0:
        return None
    return blob[i + len(MARKER): i + len(MARKER) + 32].decode("ascii", "replace")


def is_stale(path: str = OUT, extra_flags=()) -> bool:
    return embedded_hash(path) != source_hash(extra_flags)


def build(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True builds libc4a0_hip_diag.so with in-kernel phase stamps (tools/phase_profile.py) and the measured
    alternative kernels behind their environment knobs (C4_DIAG_VARIANTS: C4_TOWER32_STREAM, C4_TOWER64_HELD,
    C4_STEP_LDS_BYTES); the product library reads no tuning knob from the environment.
    The library is compiled beside its final place and renamed over it, so a failed compile never
    destroys a working one."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = OUT.replace(".so", "_diag.so") if diag else OUT
    extra = ["-DC4_PHASE_STAMPS", "-DC4_DIAG_VARIANTS"] if diag else []
    if force or is_stale(out, extra):
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [hipcc] + FLAGS + extra + ['-DC4_SOURCE_HASH="%s"' % source_hash(extra)] + SRCS + ["-o", tmp]
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed:\n" + r.stdout + r.stderr)
        os.replace(tmp, out)
        if verbose and r.stderr:
            print(r.stderr)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv))
