#!/bin/bash
# usage: tools/gemm_lab/run_lt_pair.sh [M N K [n_solutions]]   (on the GPU box)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 lt_pair.cpp -o /tmp/lt_pair -L/opt/rocm/lib -lhipblaslt
/tmp/lt_pair "$@"
