#!/bin/bash
# Per-kernel time of the numpy-callback mode (the reference's default job, trivial callback so that the tree kernels
# and the batch-building kernels are what the trace holds): gpurun -- 'bash tools/profile/callback_kernels.sh'
export TMPDIR=/tmp; O=gpurun_out/cbk; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/callback_profile.py 1700 200 > $O/run.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
cut -c1-160 $O/kernel_stats.csv | head -12
