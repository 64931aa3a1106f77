#!/bin/bash
# Per-wave phase timers of the tiled kernels (instrumented build, on the GPU
# box): the single-chain products at config 3 and the K = 2 batched products.
#   bash scripts/phase_timers.sh > gpurun_out/r03_tile_switch.txt
root=$PWD
dst=$root/gpurun_out/ab/instr
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_TILED_INSTRUMENT=1" \
   > $dst/build.log 2>&1) || { echo "instrumented build failed"; tail -5 $dst/build.log; exit 1; }
echo "== single chain (K = 1), config 3: launches 8 (X v) and 9 (X^T w)"
BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_STATS=1 BBX_TILED_DEBUG=8 python3 scripts/bench_spmv.py config3 tiled 20 2>&1 | grep -E "bbx tiled|avg"
echo "== batched (K = 2), config 3: launches 4 (X V) and 5 (X^T W)"
BBX_PACKAGE_DIR=$dst/pkg BBX_TILED_STATS=1 BBX_TILED_DEBUG=4 python3 scripts/bench_batch_products.py config3 2 10 2>&1 | grep -E "bbx tiled|avg"
for ab in 1 2 4 7; do  # (K = 1 numbers: see above run)
  echo "== ablation BBX_ABLATE=$ab (1 gathers, 2 slice loads, 4 switch barriers; wrong results, timing only)"
  BBX_PACKAGE_DIR=$dst/pkg BBX_ABLATE=$ab python3 scripts/bench_spmv.py config3 tiled 50 2>&1 | grep -E "avg"
  BBX_PACKAGE_DIR=$dst/pkg BBX_ABLATE=$ab python3 scripts/bench_batch_products.py config3 2 10 2>&1 | grep -E "avg"
done
rm -rf $dst
