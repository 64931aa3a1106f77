#!/bin/bash
# round-6 measurements, part 1 (one gpurun call)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
python3 scripts/diag/control_harness.py > $O/r06_harness.txt 2>&1
pick='import json,sys
d=json.loads(sys.stdin.readline())
r=d["repeat"]; h=d["config"].get("host") or {}
print("value %.1f it/s; blocks %s; us/cg-iter %s; n_cg %s; host %s" % (d["value"], r["values"], r["us_per_cg_iter"], r["mean_n_cg_iter"], h))'
{ for a in 1 2 4; do echo "== tiny (20k x 1k) BBX_CG_AHEAD=$a"; BBX_CG_AHEAD=$a python3 bench.py --config tiny --steps 200 --warmup 20 --burnin 50 --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"; done
  echo "== tiny base (round 5)"; BBX_PACKAGE_DIR=$PWD/ab_base/bayes-bridge_amd python3 bench.py --config tiny --steps 200 --warmup 20 --burnin 50 --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"
  for a in 1 2 4; do echo "== tiny-dense BBX_CG_AHEAD=$a"; BBX_CG_AHEAD=$a python3 bench.py --config tiny-dense --steps 200 --warmup 20 --burnin 50 --cpu-baseline-iters 0 --multi-chain 0 2>/dev/null | python3 -c "$pick"; done
} > $O/r06_cg_word_small.txt 2>&1
bash scripts/r06_tiled_counters.sh 10 > $O/r06_tiled_counters.log 2>&1
BBX_TILED_DEBUG=20 bash scripts/ab_build.sh "scripts/bench_spmv.py config2 tiled 60" "dbg|avg|geometry" "instr:-DBBX_TILED_INSTRUMENT=1" > $O/r06_config2_stamps_raw.txt 2>&1
bash scripts/rehearse_8rank.sh $O r06 > $O/r06_rehearse.log 2>&1
cat $O/r06_harness.txt $O/r06_cg_word_small.txt; tail -40 $O/r06_tiled_counters.log; tail -30 $O/r06_config2_stamps_raw.txt; tail -5 $O/r06_rehearse.log
