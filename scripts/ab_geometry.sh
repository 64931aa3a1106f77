#!/bin/bash
# A/B of tiled geometries for X^T at config 3 (env overrides, one process each)
run() { echo "== $*"; env "$@" BBX_TILED_STATS=1 python3 scripts/bench_spmv.py config3 tiled 200 2>&1 | grep -E "tdot  avg|dot   avg|50000x1000000"; }
run A=0
run BBX_TILED_PR_T=6272 BBX_TILED_BLOCKS_T=96 BBX_TILED_G_T=32
run BBX_TILED_PR_T=5056 BBX_TILED_BLOCKS_T=75 BBX_TILED_G_T=25
run BBX_TILED_PR_T=4224 BBX_TILED_BLOCKS_T=66 BBX_TILED_G_T=22
run BBX_TILED_PR_T=8192 BBX_TILED_BLOCKS_T=112 BBX_TILED_G_T=37
run BBX_TILED_PR_T=6272 BBX_TILED_BLOCKS_T=128 BBX_TILED_G_T=32
