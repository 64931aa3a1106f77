#!/bin/bash
# kernel-trace averages of the lambda / Polya-Gamma kernels with and without the
# root-based powers (see r05_lscale_pow_ab.sh)
root=$PWD
out=$root/gpurun_out/${1:-r05t}; mkdir -p $out
dst=$root/gpurun_out/ab/generic_pow
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_POS_POW_GENERIC=1" \
   > $dst/build.log 2>&1) || { echo "variant build failed"; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $root
for v in roots generic; do
  if [ $v = generic ]; then export BBX_PACKAGE_DIR=$dst/pkg; else unset BBX_PACKAGE_DIR; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$v -- python3 bench.py --steps 30 --warmup 5 --repeat 1 --cpu-baseline-iters 0 --multi-chain 0 > $out/tr_$v.json 2> $out/tr_$v.err
  f=$(find $out/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== pow=$v (name, calls, avg ns, min, max)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row.get('Name', '')
    if any(k in n for k in ('lscale', 'chain_pg', 'gscale')):
        print(n.split('(')[0][-40:], row.get('Calls'), row.get('AverageNs'), row.get('MinNs'), row.get('MaxNs'))
PY
  rm -rf $out/tr_$v
done
rm -rf $dst
