#!/bin/bash
# A/B of libbbx variants on the operator kernels: bash scripts/ab_spmv.sh <config> lib1 lib2 ...
cfg=${1:-config3}; shift
for lib in "$@"; do
  echo "=== $lib"
  BBX_LIBRARY=$PWD/bayes-bridge_amd/$lib BBX_TILED_STATS=1 timeout 600 python3 scripts/bench_spmv.py $cfg tiled 200 2>&1 | grep -E "tiled geometry|avg|max abs err|bbx tiled"
done
