#!/usr/bin/env python3
"""Builds c4a0_amd/libc4a0_hip_<name>.so from the kernel sources of a git revision (default HEAD), for same-box A/B runs:

    python tools/build_variant.py base [REV] [-DFLAG ...]     ->  c4a0_amd/libc4a0_hip_base.so      (REV "WORK" = the working tree)
    C4A0_HIP_LIB=libc4a0_hip_base.so python bench.py ...     (c4a0_amd/_lib.py loads it without the source-hash check)

Box-to-box spread is +-4 %, so a kernel change is only ever judged against the previous kernel ON THE SAME BOX, in the same
gpurun call (tools/lib_ab.sh)."""
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from c4a0_amd.csrc import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    rev = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
    defines = [a for a in sys.argv[3:] if a.startswith("-D")]
    out = os.path.join(ROOT, "c4a0_amd", f"libc4a0_hip_{name}.so")
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "c4a0_amd", "csrc"))
        os.makedirs(os.path.join(td, "include"))
        srcs = []
        for d in B.DEPS:
            rel = os.path.relpath(d, ROOT)
            data = open(d, "rb").read() if rev == "WORK" else subprocess.run(["git", "show", f"{rev}:{rel}"], cwd=ROOT, capture_output=True, check=True).stdout
            dst = os.path.join(td, rel)
            with open(dst, "wb") as f:
                f.write(data)
            if d in B.SRCS:
                srcs.append(dst)
        cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + B.FLAGS + defines + ['-DC4_SOURCE_HASH="variant-%s"' % name] + srcs + ["-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.exit(r.stdout + r.stderr)
    print(out)


if __name__ == "__main__":
    main()
