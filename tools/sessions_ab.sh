#!/bin/bash
# How many concurrent sessions (HIP graphs on their own streams) should share the resident games?  Same box, interleaved.
#   bash tools/sessions_ab.sh [rounds=2]      (the default, 2 sessions in ONE paired graph, is the first row of every round)
O=gpurun_out/sessions_ab; mkdir -p $O; : > $O/bench.txt
for r in $(seq 1 ${1:-2}); do for v in "--sessions 2" "--sessions 2 --independent-graphs" "--sessions 3" "--sessions 4"; do
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs $v $BENCH_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s %8.0f games/s  %.4f ms/round' % ('$v', d['value'], d['ms_per_round']))" | tee -a $O/bench.txt
done; done
