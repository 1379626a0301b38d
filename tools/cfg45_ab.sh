#!/bin/bash
# BASELINE configs 4 / 5 (8-block / 64-channel net) on one box: the 64-channel tower's workgroup shapes (bench.py --tower-config).
#   bash tools/cfg45_ab.sh "0 3" [rounds=2]
O=gpurun_out/cfg45_ab; mkdir -p $O; : > $O/bench.txt
for r in $(seq 1 ${2:-2}); do for t in ${1:-0 3}; do
  for shape in "--n-mcts 800 --games-per-gpu 4096" "--n-mcts 200 --games-per-gpu 8192"; do
  python bench.py --blocks 8 --channels 64 $shape --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --tower-config $t 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tower-config %-3s %-40s %8.0f games/s  %.4f ms/round' % ('$t', '$shape', d['value'], d['ms_per_round']))" | tee -a $O/bench.txt
done; done; done
