#!/bin/bash
# Why the folded direction step (bbx_design_set_cg_fold / BBX_CG_FOLD=1) does
# not pay: per-wave phase stamps of the X~ v kernel INSIDE the CG loop of a
# config-3 chain, default 4-launch loop against the 3-launch one (instrumented
# build, on the GPU box), plus the kernel-trace view of both loops.
#   bash scripts/fold_phase_timers.sh > gpurun_out/r04_cg_fold.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
root=$PWD
dst=$root/gpurun_out/ab/instr
rm -rf $dst; mkdir -p $dst
cp -r $root/bayes-bridge_amd $dst/pkg; cp -r $root/include $dst/include
(cd $dst/pkg/csrc && rm -rf build && make -j16 ../libbbx.so \
   CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -DBBX_TILED_INSTRUMENT=1" \
   > $dst/build.log 2>&1) || { echo "instrumented build failed"; tail -5 $dst/build.log; exit 1; }
for fold in 0 1; do
  echo "== BBX_CG_FOLD=$fold: phase stamps of launches 200-203 of a config-3 chain (grid 253 = X~ v, 256 = X~^T w)"
  for at in 200 202; do
    BBX_PACKAGE_DIR=$dst/pkg BBX_CG_FOLD=$fold BBX_TILED_DEBUG=$at python3 scripts/iteration_traffic.py 2 8 2>&1 | grep -E "bbx tiled dbg"
  done
done
rm -rf $dst
for fold in 0 1; do
  echo "== BBX_CG_FOLD=$fold: kernel trace of bench.py --config config3 / config2 (avg us per kernel, entry-returns excluded)"
  for cfg in config3 config2; do
    BBX_CG_FOLD=$fold rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fold_trace_$fold$cfg -- python3 bench.py --config $cfg --steps 30 --warmup 5 --burnin 100 --multi-chain 0 --cpu-baseline-iters 0 --repeat 1 > gpurun_out/fold_trace_$fold$cfg.json 2> gpurun_out/fold_trace_$fold$cfg.log
    python3 - gpurun_out/fold_trace_$fold$cfg $cfg <<'PY'
import glob, json, sys
import pandas as pd
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
t = pd.read_csv(f)
t["dur"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
t["name"] = t.Kernel_Name.map(lambda s: s.split("(")[0].replace("void ", "").replace("bbx::", "")[:60])
t = t[t.name.str.contains("tiled_spmv|tdot_finalize|cg_direction")]
t = t.sort_values("Start_Timestamp")
rows = []
for (name, g), sel in t.groupby(["name", "Grid_Size_X"]):
    keep = sel[sel.dur >= .5 * sel.dur.median()]
    rows.append((keep.dur.sum(), "%-52s grid=%-4d launches %5d (+%d at entry)  avg %7.2f us" % (
        name, g // sel.Workgroup_Size_X.iloc[0], len(keep), len(sel) - len(keep), keep.dur.mean())))
for _, r in sorted(rows, reverse=True)[:6]:
    print("  " + r)
line = json.loads(open(sys.argv[1] + ".json").read().strip().splitlines()[-1])
print("  %s: %.1f Gibbs it/s under the tracer, n_cg %.1f, %d launches per CG iteration" % (
    sys.argv[2], line["value"], line["config"]["mean_n_cg_iter"], line["config"]["cg_launches_per_iteration"]))
PY
    rm -rf gpurun_out/fold_trace_$fold$cfg gpurun_out/fold_trace_$fold$cfg.json gpurun_out/fold_trace_$fold$cfg.log
  done
done
