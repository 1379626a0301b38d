"""The library's host-side result functions under AddressSanitizer + UBSan (CPU build; GPU sanitizers are not available on the
pool): `pickle.loads` feeds c4_cbor_to_records bytes from a file, so the decoder is fuzzed with damaged documents and every
buffer it touches is an exact-size heap allocation (tests/sanitize_results_host.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_codec_and_shuffle_under_address_and_ub_sanitizers(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "sanitize_results_host")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
           "-x", "c++", os.path.join(ROOT, "c4a0_amd", "csrc", "c4_results_host.hip"), os.path.join(ROOT, "tests", "sanitize_results_host.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "ubsan" in r.stderr.lower()) and "cannot find" in r.stderr:
        pytest.skip("sanitizer runtimes not installed")
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert r.returncode == 0 and r.stdout.startswith("sanitized ok"), (r.stdout[-500:], r.stderr[-3000:])
    n_ok, n_bad = [int(w) for w in r.stdout.split() if w.isdigit()][1:3]
    assert n_bad > 100000 and n_ok > 1000, r.stdout      # both sides of the decoder were exercised
