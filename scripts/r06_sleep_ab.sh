#!/bin/bash
# The host between two stop tests: poll throughout (BBX_CG_SLEEP=0) or sleep
# through 3/4 of a long iteration (default).  Speed and CPU seconds per step.
pick='import json,sys
d=json.loads(sys.stdin.readline())
r=d["repeat"]; h=d["config"].get("host") or {}
print("value %.1f it/s; us/cg-iter %s; host %s" % (d["value"], r["us_per_cg_iter"], h))'
for rep in 1 2; do
for s in 1 0; do
  echo "== config3 BBX_CG_SLEEP=$s"
  BBX_CG_SLEEP=$s python3 bench.py --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"
done
done
echo "== config3 --cg-fold 1 (3-launch iteration)"
python3 bench.py --cpu-baseline-iters 0 --multi-chain 0 --cg-fold 1 2>/dev/null | python3 -c "$pick"
echo "== config2 (AHEAD 1, period 35 us: no sleep)"
python3 bench.py --config config2 --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"
