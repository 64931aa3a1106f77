#!/bin/bash
# A/B: the tilted-stable sampler's powers as roots / small integer powers for
# a = 1/4 (samplers.hpp pos_pow) against exp(y log x) everywhere
# (-DBBX_POS_POW_GENERIC build in a scratch copy of the package)
root=$PWD
out=$root/gpurun_out/${1:-r05s}; mkdir -p $out
dst=$root/gpurun_out/ab/generic_pow
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_POS_POW_GENERIC=1" \
   > $dst/build.log 2>&1) || { echo "variant build failed"; tail -5 $dst/build.log; exit 1; }
for cfg in config3 config2; do for rep in 1 2 3; do for v in roots generic; do
  if [ $v = generic ]; then export BBX_PACKAGE_DIR=$dst/pkg; else unset BBX_PACKAGE_DIR; fi
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/pw_${cfg}_${v}_$rep.json 2> $out/pw_${cfg}_${v}_$rep.err
  python3 -c "import json;d=json.load(open('$out/pw_${cfg}_${v}_$rep.json'));print('$cfg pow=$v rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
unset BBX_PACKAGE_DIR
cd /tmp && export TMPDIR=/tmp && cd $root
for v in roots generic; do
  if [ $v = generic ]; then export BBX_PACKAGE_DIR=$dst/pkg; else unset BBX_PACKAGE_DIR; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$v -- python3 bench.py --steps 30 --warmup 5 --repeat 1 --cpu-baseline-iters 0 --multi-chain 0 > $out/tr_$v.json 2> $out/tr_$v.err
  f=$(ls $out/tr_$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== kernel stats, pow=$v"; [ -n "$f" ] && grep -E "lscale|pg_kernel|gscale" "$f" | cut -c1-110
  rm -rf $out/tr_$v
done
rm -rf $dst
