#!/bin/bash
# A/B of libbbx build variants inside ONE gpurun call:
#   bash scripts/ab_build.sh "<python script + args>" "<grep pattern>" "name1:<extra hipcc flags>" ...
# Every variant is built under gpurun_out/ab/<name>/ as a private copy of the
# package (nothing but the three product libraries lives in bayes-bridge_amd/),
# and the script imports that copy through BBX_PACKAGE_DIR.
cmd=$1; pat=$2; shift 2
root=$PWD
for spec in "base:" "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  dst=$root/gpurun_out/ab/$name
  rm -rf $dst; mkdir -p $dst
  cp -r $root/bayes-bridge_amd $dst/pkg
  cp -r $root/include $dst/include 2>/dev/null
  if [ -n "$flags" ]; then
    (cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
       CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flags" \
       LAYOUT_DEFS="$flags" \
       > $dst/build.log 2>&1) || { echo "build of $name failed"; tail -5 $dst/build.log; continue; }
  fi
  echo "=== $name ($flags)"
  BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_STATS=1 timeout 900 python3 $cmd 2>&1 | grep -E "$pat"
  rm -rf $dst   # nothing of a variant travels back with gpurun_out/
done
