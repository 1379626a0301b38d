# The chip partitioned between the two sessions by queue CU masks, in the library's host loop (diagnostic build) -> profiles/r06_cu_partition.txt
python c4a0_amd/csrc/build.py --diag > /dev/null
mkdir -p gpurun_out/r6u; : > gpurun_out/r6u/cu_mask.txt
export C4A0_HIP_LIB=libc4a0_hip_diag.so
run() { echo "== $1" >> gpurun_out/r6u/cu_mask.txt; env $2 timeout 200 python tools/whole_call.py 40960 4096 --reps 2 --host-loop native 2>>gpurun_out/r6u/err.txt | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); p = d['phases']
    print('%.4f s  %.1f games/s  steady %.4f s over %s rounds = %.2f us/round  tail %.4f' % (d['seconds'], d['games_per_s'], p['steady_s'], p['rounds_until_all_started'], p['steady_s'] / max(1, p['rounds_until_all_started']) * 1e6, p['tail_s']))" >> gpurun_out/r6u/cu_mask.txt; }
run "graph (the product schedule), diagnostic library" "C4_X=0"
run "eager launches, no CU masks" "C4_PAIR_EAGER=1"
run "eager launches, 16 CUs of every XCD per session" "C4_PAIR_EAGER=1 C4_PAIR_CU_MASK=1"
run "eager launches, four whole XCDs per session" "C4_PAIR_EAGER=1 C4_PAIR_CU_MASK=2"
cat gpurun_out/r6u/cu_mask.txt; tail -3 gpurun_out/r6u/err.txt
