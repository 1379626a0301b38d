mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_mcts_parity.py tests/test_gpu_play_games.py tests/test_golden_selfplay.py tests/test_gpu_full_size.py -m gpu -x -q > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f/pytest.log
python tools/tree_roofline.py --games 2048,4096,16384,65536,131072 > gpurun_out/r02f/sweep_w4.json 2> gpurun_out/r02f/sweep_w4.err
for w in 3 5; do C4A0_HIP_LIB=libc4a0_hip_w$w.so python tools/tree_roofline.py --games 2048,65536,131072 > gpurun_out/r02f/sweep_w$w.json 2> gpurun_out/r02f/sweep_w$w.err; done
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r02f/bench.json 2> gpurun_out/r02f/bench.err
python tools/callback_mode_rate.py 16384 > gpurun_out/r02f/callback.txt 2>&1
