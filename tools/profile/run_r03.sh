#!/bin/bash
# Round-3 evidence for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools/profile/run_r03.sh').
O=gpurun_out/r03p; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --rounds-per-step 64 --preroll 640 --instrumented-steps 300 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_step_kernel 800 > $O/traffic_$c.json
  rm -rf $O/pmc_$c
done
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --independent-graphs > $O/bench_independent_graphs.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gemm hipblaslt > $O/bench_hipblaslt.json 2>/dev/null
python bench.py --whole-job > $O/whole_job.json 2> $O/whole_job.err
python tools/tree_roofline.py --games 2048,4096,16384,65536,131072 > $O/tree_sweep.json 2> $O/tree_sweep.err
bash tools/tower_ab.sh "2048 4096" > $O/tower.txt 2>&1
python tools/tower_probe.py 64 8 2048 >> $O/tower.txt 2>&1
python tools/callback_mode_rate.py 16384 > $O/callback_mode.txt 2>&1
python tools/callback_breakdown.py 1700 1400 2>&1 | grep '^run' >> $O/callback_mode.txt
python tools/callback_breakdown.py 8192 100 4 32 4 2 2>&1 | grep '^run' >> $O/callback_mode.txt
bash tools/profile/whole_job_kernels.sh > $O/whole_job_kernels.txt 2>&1      # -> gpurun_out/wj/kernel_stats.csv
bash tools/profile/callback_kernels.sh > $O/callback_kernels.txt 2>&1        # -> gpurun_out/cbk/kernel_stats.csv
# counters under the evaluator (ten --pmc passes per backend, ~2 minutes): bash tools/profile/run_r03_eval_pmc.sh 2048 r03
ls -la $O
