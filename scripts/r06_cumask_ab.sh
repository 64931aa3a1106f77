#!/bin/bash
# Experiment: the tau / lambda branch's stream confined to n CUs (every k-th).
pick='import json,sys
d=json.loads(sys.stdin.readline())
r=d["repeat"]
print("value %.1f it/s; median block %.1f; us/cg-iter %s" % (d["value"], r["median"], r["us_per_cg_iter"]))'
for rep in 1 2 3 4; do
for e in "BBX_NOP=1" "BBX_BRANCH_CUS=64 BBX_BRANCH_CU_STRIDE=4" "BBX_BRANCH_CUS=128 BBX_BRANCH_CU_STRIDE=2" "BBX_BRANCH_CUS=32 BBX_BRANCH_CU_STRIDE=8"; do
  echo "== config3 $e"
  env $e python3 bench.py --cpu-baseline-iters 0 --multi-chain 0 --live-traffic 0 2>/dev/null | python3 -c "$pick"
done
done
