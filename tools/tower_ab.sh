#!/bin/bash
# The conv tower's workgroup shapes (c4_conv_tower_bf16's config: 1 = 16 boards, 2 = 8 boards, 3 = 16 boards on 12
# wavefronts) on one box: us per launch.
for n in ${1:-2048 4096}; do for v in 1 3 2; do
  echo -n "config $v: "; python tools/tower_probe.py 32 4 $n $v 2>&1 | tail -1
done; done
