#!/bin/bash
# Round-2 evidence for profiles/: run on the GPU box from the repo root (gpurun -- 'bash tools/profile/run_r02.sh').
O=gpurun_out/r02p; mkdir -p $O; export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 bench.py --steps 1 --warmup 1 --rounds-per-step 64 --preroll 640 --instrumented-steps 300 --no-cpu-baseline > $O/pmc_$c.json 2> $O/pmc_$c.err
  python tools/profile/summarize_pmc.py $O/pmc_$c c4_step_kernel 800 > $O/traffic_$c.json
done
python tools/tree_roofline.py --games 2048,4096,16384,65536,131072 > $O/tree_sweep.json 2> $O/tree_sweep.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq65536 -- python3 tools/tree_roofline.py --games 65536 --steps 40 --preroll 1500 > /dev/null 2>&1
python tools/profile/summarize_pmc.py $O/sq65536 c4_step_kernel 1500 > $O/sq_65536.json
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/sq2048 -- python3 tools/tree_roofline.py --games 2048 --steps 40 --preroll 1500 > /dev/null 2>&1
python tools/profile/summarize_pmc.py $O/sq2048 c4_step_kernel 1500 > $O/sq_2048.json
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sqb65536 -- python3 tools/tree_roofline.py --games 65536 --steps 40 --preroll 1500 > /dev/null 2>&1
python tools/profile/summarize_pmc.py $O/sqb65536 c4_step_kernel 1500 > $O/sq_active_65536.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/t65536_$c -- python3 tools/tree_roofline.py --games 65536 --steps 40 --preroll 1500 > /dev/null 2>&1
  python tools/profile/summarize_pmc.py $O/t65536_$c c4_step_kernel 1500 > $O/traffic65536_$c.json
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_lab tools/gather_lab.hip && /tmp/gather_lab > $O/gather_lab.jsonl 2> $O/gather_lab.err
python bench.py --whole-job > $O/whole_job.json 2> $O/whole_job.err
python tools/callback_mode_rate.py 16384 > $O/callback_mode.txt 2>&1
bash tools/occupancy_probe.sh > /dev/null 2>&1; cp gpurun_out/occ/occupancy.txt $O/occupancy.txt
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/sq65536 $O/sq2048 $O/sqb65536 $O/t65536_FETCH_SIZE $O/t65536_WRITE_SIZE
ls -la $O
