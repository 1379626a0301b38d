#!/bin/bash
# The FETCH_SIZE / WRITE_SIZE passes of tools/profile/run_r06.sh alone (each --pmc pass its own run, --kernel-trace only), re-run whenever
# c4_session.hip or c4_device.hpp changes: profiles/step_kernel_traffic.json records the hash of those two files, and bench.py says whether
# the traffic figure was collected on the step kernel it is running.   gpurun -- 'bash tools/profile/run_r06_traffic.sh'
O=gpurun_out/r06t; mkdir -p $O; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --rounds-per-step 64 --preroll 640 --instrumented-steps 300 --no-cpu-baseline --no-other-configs > $O/pmc_$c.json 2> $O/pmc_$c.err
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_step_kernel 20 > $O/traffic_$c.json
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_out_step_kernel -300 > $O/traffic_fused_$c.json
  rm -rf $O/pmc_$c
done
cat $O/traffic_FETCH_SIZE.json $O/traffic_WRITE_SIZE.json $O/traffic_fused_FETCH_SIZE.json $O/traffic_fused_WRITE_SIZE.json
