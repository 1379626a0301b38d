#!/bin/bash
# The paired graph's schedule as the device ran it: kernel start / end stamps of a short bench run (rocprofv3 --kernel-trace),
# reduced by tools/profile/pair_timeline.py to one table per steady-state round.   gpurun -- 'bash tools/profile/pair_timeline.sh'
O=gpurun_out/pair_tl; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err
f=$(find $O/trace -name '*kernel_trace.csv' | head -1)
python tools/profile/pair_timeline.py $f > $O/timeline.txt
head -c 3000000 $f > $O/kernel_trace_head.csv
rm -rf $O/trace
tail -60 $O/timeline.txt
