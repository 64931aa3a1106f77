#!/bin/bash
# A/B of libbbx build variants on the operator kernels, inside ONE gpurun call:
#   bash scripts/ab_spmv.sh <config> "name1:<extra hipcc flags>" "name2:..." ...
# Every variant is built under gpurun_out/ab/<name>/ as a private copy of the
# package (nothing but the three product libraries lives in bayes-bridge_amd/),
# and the bench script imports that copy through PYTHONPATH.
cfg=${1:-config3}; shift
root=$PWD
for spec in "base:" "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  dst=$root/gpurun_out/ab/$name
  rm -rf $dst; mkdir -p $dst
  cp -r $root/bayes-bridge_amd $dst/pkg
  cp -r $root/include $dst/include 2>/dev/null
  if [ -n "$flags" ]; then
    (cd $dst/pkg/csrc && rm -rf build && make -j8 ../libbbx.so \
       CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -I$root $flags" \
       > $dst/build.log 2>&1) || { echo "build of $name failed"; tail -5 $dst/build.log; continue; }
  fi
  echo "=== $name ($flags)"
  BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_STATS=1 timeout 600 python3 scripts/bench_spmv.py $cfg tiled 200 2>&1 \
    | grep -E "tiled geometry|avg|max abs err|bbx tiled"
done
