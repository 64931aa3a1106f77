#!/bin/bash
# Generic A/B on the GPU box: the working tree's library against a copy of the
# package kept under gpurun_out/ab_ref (made on the build host by
#   rm -rf gpurun_out_ref && mkdir -p ab_ref && git archive <rev> bayes-bridge_amd include | tar -x -C ab_ref
# -- gpurun_out/ does not travel, so the reference copy lives in ./ab_ref, which
# is git-ignored through .git/info/exclude).  Usage: r05_ab_against_ref.sh <out> <configs...>
root=$PWD
out=$root/gpurun_out/${1:-r05v}; shift; mkdir -p $out
ref=$root/ab_ref
(cd $ref/bayes-bridge_amd/csrc && rm -rf build && make -j16 ../libbbx.so > $out/ref_build.log 2>&1) \
  || { echo "reference build failed"; tail -5 $out/ref_build.log; exit 1; }
cp $root/bayes-bridge_amd/libbbx_hostrng.so $root/bayes-bridge_amd/libbbx_layout.so $ref/bayes-bridge_amd/ 2>/dev/null
for cfg in "$@"; do for rep in 1 2 3; do for v in work ref; do
  if [ $v = ref ]; then export BBX_PACKAGE_DIR=$ref/bayes-bridge_amd; else unset BBX_PACKAGE_DIR; fi
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/ab_${cfg}_${v}_$rep.json 2> $out/ab_${cfg}_${v}_$rep.err
  python3 -c "import json;d=json.load(open('$out/ab_${cfg}_${v}_$rep.json'));print('$cfg $v rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
