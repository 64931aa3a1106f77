#!/bin/bash
# Round 5: early slice fill (BBX_TILED_EARLY_FILL=1, default) against the fill at
# the tile switch (=0): production build timing, then the instrumented build's
# per-wave phase timers.
for e in 0 1; do
  echo "== production build, BBX_TILED_EARLY_FILL=$e"
  BBX_TILED_EARLY_FILL=$e timeout 300 python3 scripts/bench_spmv.py config3 tiled 200 2>&1 | grep -E "avg|err"
done
root=$PWD
dst=$root/gpurun_out/ab/instr
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_TILED_INSTRUMENT=1" \
   > $dst/build.log 2>&1) || { echo "instrumented build failed"; tail -5 $dst/build.log; exit 1; }
for e in 0 1; do
  echo "== instrumented build, BBX_TILED_EARLY_FILL=$e: launches 8 (X v) and 9 (X^T w)"
  BBX_TILED_EARLY_FILL=$e BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_DEBUG=8 timeout 300 python3 scripts/bench_spmv.py config3 tiled 20 2>&1 | grep -E "bbx tiled dbg|avg|err"
done
rm -rf $dst
