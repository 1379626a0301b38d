#!/bin/bash
# the automatic GEMM tile up to 1 024 rows picks the loader-wave forms (41 / 42 / 44 / 43), which were measured with each GEMM
# ALONE.  Same box, the PAIRED graph (two sessions sharing the chip) at 512 and 1 024 rows per session: loader-wave forms against round 3's
# tiles (27 / 9 / 23 / 10: bench.py --no-loader-waves).   bash tools/small_rows_ab.sh [rounds=2]
O=gpurun_out/small_rows_ab; mkdir -p $O; : > $O/bench.txt
for r in $(seq 1 ${1:-2}); do for g in 1024 2048; do for v in "" "--no-loader-waves"; do
  python bench.py --games-per-gpu $g --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs $v 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%5d games (2 x %4d rows) %-20s %8.0f games/s  %.4f ms/round' % ($g, $g // 2, '$v' or 'loader-wave tiles', d['value'], d['ms_per_round']))" | tee -a $O/bench.txt
done; done; done
