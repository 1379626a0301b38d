#!/bin/bash
# Same-box A/B of kernel builds: for each library name (c4a0_amd/libc4a0_hip_<name>.so; "product" = libc4a0_hip.so) the head
# GEMMs alone (tools/gemm_probe.py) and the bench (games/s, ms per round), alternating so that drift hits all alike.
#   bash tools/lib_ab.sh "base product" [rounds=2] [probe-cfgs=11]      (BENCH_ARGS="..." adds arguments to every bench run;
#   a name of the form product:ARGS runs the product library with extra bench arguments, e.g. "product:--gemm-write-through")
O=gpurun_out/lib_ab; mkdir -p $O; : > $O/bench.txt
for r in $(seq 1 ${2:-2}); do for n in $1; do
  extra=""; lib=$n
  case $n in *:*) lib=${n%%:*}; extra=${n#*:};; esac
  if [ $lib = product ]; then unset C4A0_HIP_LIB; else export C4A0_HIP_LIB=libc4a0_hip_$lib.so; fi
  [ $r = 1 ] && { echo "== $n"; PROBE_ONLY=alone PROBE_CFGS=${3:-11} python tools/gemm_probe.py 2048 2>&1 | grep "alone:"; } | tee -a $O/probe.txt
  python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs $BENCH_ARGS $extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-32s %8.0f games/s  %.4f ms/round' % ('$n', d['value'], d['ms_per_round']))" | tee -a $O/bench.txt
done; done
