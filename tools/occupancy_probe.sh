#!/bin/bash
# How the step kernel's launch time scales with the wavefronts a SIMD may hold (same binary: unused
# dynamic LDS caps the workgroups per CU).  12 per CU = 3 per SIMD is what the registers allow.
# Needs the diagnostic build (python c4a0_amd/csrc/build.py --diag): the product library reads no environment knob.
export C4A0_HIP_LIB=libc4a0_hip_diag.so
mkdir -p gpurun_out/occ
for lds in 0 16384 20480 40960 81920; do
  C4_STEP_LDS_BYTES=$lds python tools/tree_roofline.py --games 2048,65536 --steps 100 --preroll 1500 2>&1 >/dev/null | grep '^{' | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print('lds_pad=$lds games=%d device_clock_us=%.1f' % (r['games_per_launch'], r['device_clock_us']))
"
done | tee gpurun_out/occ/occupancy.txt
