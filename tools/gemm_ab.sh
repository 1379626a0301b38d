#!/bin/bash
# A/B of head-GEMM tile configurations on ONE box: alone (tools/gemm_probe.py) and in the bench (games/s, ms per round).
#   bash tools/gemm_ab.sh "11,2,29" "0 29 29,11"      (probe configs; bench --gemm-config values)
O=gpurun_out/gemm_ab; mkdir -p $O
PROBE_ONLY=alone PROBE_CFGS=${1:-11,29} python tools/gemm_probe.py 2048 2>&1 | grep "alone:" | tee $O/probe.txt
for c in ${2:-0}; do
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs --gemm-config $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('gemm-config %-8s %8.0f games/s  %.4f ms/round' % ('$c', d['value'], d['ms_per_round']))" | tee -a $O/bench.txt
done
