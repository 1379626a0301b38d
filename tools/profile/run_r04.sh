#!/bin/bash
# Round-4 evidence for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools/profile/run_r04.sh').
O=gpurun_out/r04p; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --rounds-per-step 64 --preroll 640 --instrumented-steps 300 --no-cpu-baseline --no-other-configs > $O/pmc_$c.json 2> $O/pmc_$c.err
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_step_kernel 20 > $O/traffic_$c.json   # the 300 instrumented launches (the timed region runs the step inside c4_out_step_kernel)
  rm -rf $O/pmc_$c
done
python tools/tree_roofline.py --games 2048,4096,16384,65536,131072 > $O/tree_sweep.json 2> $O/tree_sweep.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wj -- python3 bench.py --whole-job --whole-job-modes device_mode > $O/whole_job_under_rocprof.json 2> $O/whole_job_under_rocprof.err
cp $(find $O/wj -name '*kernel_stats.csv' | head -1) $O/whole_job_kernel_stats.csv; rm -rf $O/wj
# counters under the evaluator (ten --pmc passes per backend, ~2 minutes): bash tools/profile/run_r03_eval_pmc.sh 2048 r04
ls -la $O
