#!/bin/bash
# Round-6 evidence for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools/profile/run_r06.sh').
# Every step is bounded by its own timeout; the summaries land under gpurun_out/r06p and are copied into profiles/ by hand.
# Counter passes (--pmc) are their own runs, with --kernel-trace only (never with --stats / other trace domains).
O=gpurun_out/r06p; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python bench.py --steps 20 --warmup 1 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
# HBM traffic of the tree kernel: FETCH_SIZE and WRITE_SIZE, one pass each, of the bench command; summarised for the stand-alone
# c4_step_kernel (the 300 instrumented launches) and for c4_out_step_kernel (the last 300 launches = the event-bracketed rounds)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --rounds-per-step 64 --preroll 640 --instrumented-steps 300 --no-cpu-baseline --no-other-configs > $O/pmc_$c.json 2> $O/pmc_$c.err
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_step_kernel 20 > $O/traffic_$c.json
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_out_step_kernel -300 > $O/traffic_fused_$c.json
  rm -rf $O/pmc_$c
done
timeout 300 python tools/tree_roofline.py --games 2048,4096,16384,65536,131072 > $O/tree_sweep.json 2> $O/tree_sweep.err
# the step kernel's instruction and issue counters on the CURRENT kernel (two passes of 8 counters each, per launch size)
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY"
SQ2="SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for g in 65536 2048; do
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $O/sq$g -- python3 tools/tree_roofline.py --games $g --steps 40 --preroll 1500 > /dev/null 2>&1
  python tools/profile/summarize_pmc.py $O/sq$g c4_step_kernel 1500 > $O/sq_$g.json; rm -rf $O/sq$g
  timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $O/sqb$g -- python3 tools/tree_roofline.py --games $g --steps 40 --preroll 1500 > /dev/null 2>&1
  python tools/profile/summarize_pmc.py $O/sqb$g c4_step_kernel 1500 > $O/sq_active_$g.json; rm -rf $O/sqb$g
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t65536_$c -- python3 tools/tree_roofline.py --games 65536 --steps 40 --preroll 1500 > /dev/null 2>&1
  python tools/profile/summarize_pmc.py $O/t65536_$c c4_step_kernel 1500 > $O/traffic65536_$c.json; rm -rf $O/t65536_$c
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wj -- python3 bench.py --whole-job --whole-job-modes device_mode > $O/whole_job_under_rocprof.json 2> $O/whole_job_under_rocprof.err
cp $(find $O/wj -name '*kernel_stats.csv' | head -1) $O/whole_job_kernel_stats.csv; rm -rf $O/wj
ls -la $O
