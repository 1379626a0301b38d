O=gpurun_out/r06f; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 1 --no-cpu-baseline --no-other-configs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv; rm -rf $O/stats
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wj -- python3 bench.py --whole-job --whole-job-modes device_mode > $O/whole_job_under_rocprof.json 2> $O/whole_job_under_rocprof.err
cp $(find $O/wj -name '*kernel_stats.csv' | head -1) $O/whole_job_kernel_stats.csv; rm -rf $O/wj
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
tail -c 600 $O/bench.json; tail -2 $O/smoke.txt; ls -la $O
