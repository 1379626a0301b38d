#!/usr/bin/env python3
"""The shader clock the head GEMMs' main loops get INSIDE the bench (two sessions, everything running): the diagnostic
build's stamps (c4_debug_gemm_clock) read after bench.py's own main() has run in this process.
    python tools/bench_clock.py [bench.py arguments]        (needs python tools/build_variant.py clk WORK -DC4_GEMM_CLOCK: the product kernels + the GEMM's clock stamps)"""
import ctypes as C
import os
import sys

os.environ.setdefault("C4A0_HIP_LIB", "libc4a0_hip_clk.so")   # python tools/build_variant.py clk WORK -DC4_GEMM_CLOCK
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from c4a0_amd import _lib  # noqa: E402

sys.argv = ["bench.py", "--no-cpu-baseline", "--no-other-configs"] + sys.argv[1:]
bench.main()
L = _lib.lib()
L.c4_debug_gemm_clock.restype = C.c_int
L.c4_debug_gemm_clock.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
g, u, n = C.c_double(), C.c_double(), C.c_uint64()
_lib.check(L.c4_debug_gemm_clock(C.byref(g), C.byref(u), C.byref(n), 1))
sys.stderr.write(f"GEMM main loops over the whole run: {u.value:.2f} us per workgroup at {g.value:.3f} GHz = {u.value * g.value * 1e3 / 21:.0f} cycles per k-tile ({n.value} workgroups)\n")
