mkdir -p gpurun_out/r02i
python -m pytest tests/test_gpu_play_games.py -m gpu -x -q > gpurun_out/r02i/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02i/pytest.log
hipcc --offload-arch=gfx950 -O3 -o /tmp/gather_lab tools/gather_lab.hip && /tmp/gather_lab 1 > gpurun_out/r02i/gather_lab.jsonl 2> gpurun_out/r02i/gather_lab.err
python tools/callback_mode_rate.py 16384 > gpurun_out/r02i/callback.txt 2>&1
for m in 2048 4096; do python tools/gemm_probe.py $m > gpurun_out/r02i/gemm_$m.txt 2>&1; PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tune_$m.csv python tools/gemm_probe.py $m > gpurun_out/r02i/gemm_tuned_$m.txt 2>&1; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02i/bench_default.json 2> gpurun_out/r02i/bench_default.err
PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_FILENAME=/tmp/tune_bench.csv python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02i/bench_tuned.json 2> gpurun_out/r02i/bench_tuned.err
cp /tmp/tune_bench*.csv gpurun_out/r02i/ 2>/dev/null
python bench.py --whole-job > gpurun_out/r02i/whole_job.json 2> gpurun_out/r02i/whole_job.err
