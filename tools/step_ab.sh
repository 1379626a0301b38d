#!/bin/bash
# A/B of step-kernel builds on ONE box: tools/tree_roofline.py (uniform evaluator, device clock) per library.
#   bash tools/step_ab.sh "libc4a0_hip_old.so libc4a0_hip.so libc4a0_hip_w3.so" 2048,4096,65536
LIBS=${1:-"libc4a0_hip.so"}; GAMES=${2:-2048,4096,65536}; O=gpurun_out/step_ab; mkdir -p $O
for rep in 1 2; do
for L in $LIBS; do
  C4A0_HIP_LIB=$L python tools/tree_roofline.py --games $GAMES --steps 300 > $O/${L%.so}_$rep.json 2> $O/${L%.so}_$rep.err
  python -c "
import json
for r in json.load(open('$O/${L%.so}_$rep.json'))['sweep']:
    print('$L rep$rep', r['games_per_launch'], 'device_us', round(r['device_clock_us'], 2), 'event_us', round(r['event_us'], 2), 'frac_dev', round(r['frac_device_clock'], 4))
"
done; done
