#!/bin/bash
# A/B of the CG loop's host side (round 6): the look through hipMemcpyAsync +
# event wait of rounds 2-5 (package built from the round-5 HEAD under ab_base/)
# against the host-mapped progress word, at BBX_CG_AHEAD = 1, 2, 3.
#   (before: rm -rf ab_base && mkdir ab_base && git archive ef6ce49 bayes-bridge_amd include | tar -x -C ab_base
#    && make -C ab_base/bayes-bridge_amd/csrc -j8; ab_base/ is not kept in the tree)
#   bash scripts/r06_cg_word_ab.sh > gpurun_out/r06_cg_word_ab.txt
pick='import json,sys
d=json.loads(sys.stdin.readline())
r=d["repeat"]; h=d["config"].get("host") or {}
print("value %.1f it/s; blocks %s; us/cg-iter %s; n_cg %s; host %s" % (d["value"], r["values"], r["us_per_cg_iter"], r["mean_n_cg_iter"], h))'
for cfg in config3 config2; do
  for rep in 1 2; do
    echo "== $cfg base (round-5 look: memcpy + event), run $rep"
    BBX_PACKAGE_DIR=$PWD/ab_base/bayes-bridge_amd python3 bench.py --config $cfg --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"
    for a in 1 2 3; do
      echo "== $cfg progress word, BBX_CG_AHEAD=$a, run $rep"
      BBX_CG_AHEAD=$a python3 bench.py --config $cfg --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"
    done
  done
done
