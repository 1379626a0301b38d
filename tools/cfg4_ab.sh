#!/bin/bash
# BASELINE config 4 / config 5's share (8 x 64 net) by head-GEMM tile configuration, same box:  bash tools/cfg4_ab.sh "0 7,11 6,11" [5]
for c in $1; do
  if [ "${2:-4}" = 5 ]; then A="--n-mcts 200 --games-per-gpu 8192"; else A="--n-mcts 800 --games-per-gpu 4096"; fi
  python bench.py --blocks 8 --channels 64 $A --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --gemm-config $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config ${2:-4} gemm-config %-8s %8.0f games/s  %.4f ms/round' % ('$c', d['value'], d['ms_per_round']))"
done
