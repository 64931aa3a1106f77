#!/bin/bash
# Per-wave phase timers (instrumented build) of the tiled products at config 2:
# where do 13 us go when a workgroup streams 133 KB?
root=$PWD
dst=$root/gpurun_out/ab/instr
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_TILED_INSTRUMENT=1" \
   > $dst/build.log 2>&1) || { echo "instrumented build failed"; tail -5 $dst/build.log; exit 1; }
for cfg in config2 config3; do
  echo "== $cfg: launches 8 (X v) and 9 (X^T w), instrumented build"
  BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_STATS=1 BBX_TILED_DEBUG=8 timeout 300 python3 scripts/bench_spmv.py $cfg tiled 20 2>&1 | grep -E "bbx tiled|avg"
done
rm -rf $dst
