#!/bin/bash
# Small-batch head GEMMs alone (latency mode): tile configurations vs their wave-specialised forms, us per launch.
for M in ${1:-256 512 1024 1700 2048}; do
  PROBE_ONLY=alone PROBE_CFGS=${2:-27,41,45,9,42,23,40,44,10,43,11,35} python tools/gemm_probe.py $M 2>&1 | grep "alone:"
done
