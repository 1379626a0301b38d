#!/bin/bash
# The conv tower's experimental variants (C4_TOWER_VARIANT, c4_conv_tower.hip) on one box: us per launch.
for n in ${1:-2048 4096}; do for v in 0 1 7; do
  echo -n "variant $v: "; C4_TOWER_VARIANT=$v python tools/tower_probe.py 32 4 $n 2>&1 | tail -1
done; done
