#!/bin/bash
# config 2 (100k x 10k): does a launch that fills all 256 CUs pay?  The geometry
# search steps PR by 128 rows: X gets PR = 512 (196 workgroups), X^T PR = 384
# (27 panels x 7 groups = 189).  BBX_TILED_PR / _PR_T force other heights.
pick='import json,sys
d=json.loads(sys.stdin.readline())
r=d["repeat"]; ro=d["roofline"]["other"]
print("value %.1f it/s; us/cg-iter %s; grids %s; dot %.2f us tdot %.2f us" % (d["value"], r["us_per_cg_iter"], d["config"]["launch_grids"], 1e3*ro["dot"]["avg_ms"], 1e3*ro["tdot"]["avg_ms"]))'
run() { echo "== $1"; env $1 python3 bench.py --config config2 --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"; }
for rep in 1 2; do
run "BBX_NOP=1"
run "BBX_TILED_PR=392"
run "BBX_TILED_PR=392 BBX_TILED_PR_T=280"
run "BBX_TILED_PR=392 BBX_TILED_PR_T=320"
run "BBX_TILED_PR=448 BBX_TILED_PR_T=320"
done
