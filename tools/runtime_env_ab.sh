#!/bin/bash
# ROCm runtime knobs against the bench (same box, alternating): none moves the paired graph except GPU_MAX_HW_QUEUES, and that one only
# downwards (8 queues: 11.2 k games/s -- the graph's nodes spread over more hardware queues and every edge becomes a cross-queue wait).
run() { python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %8.0f games/s  %.4f ms/round' % ('$1', d['value'], d['ms_per_round']))"; }
for r in 1 2; do
run default
for kv in ${@:-DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 HIP_FORCE_DEV_KERNARG=1 HIP_FORCE_DEV_KERNARG=0 GPU_MAX_HW_QUEUES=2 GPU_MAX_HW_QUEUES=8 HSA_ENABLE_INTERRUPT=0 ROC_SIGNAL_POOL_SIZE=512}; do
  env $kv true; export $kv; run $kv; unset ${kv%%=*}
done
done
