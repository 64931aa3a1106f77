#!/bin/bash
# Round-3 profiles, one gpurun call (kernel traces and --pmc passes are always
# separate rocprofv3 runs).  Writes under gpurun_out/; condensed into profiles/
# by scripts/summarize_profiles.py r03 and by hand (see profiles/README.md).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
# plain bench lines
python3 bench.py > $O/r03_bench.json 2> $O/r03_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r03_bench_driverflags.json 2>> $O/r03_bench.err
python3 bench.py --config config2 > $O/r03_bench_config2.json 2>> $O/r03_bench.err
python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain-steps 6 > $O/r03_bench_config4.json 2>> $O/r03_bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-iters 0 > $O/r03_bench_1rank_rccl.json 2>> $O/r03_bench.err
# kernel traces of the same commands
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_trace -- python3 bench.py --cpu-baseline-iters 0 > $O/r03_bench_under_rocprof.json 2> $O/r03_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r03_dense_trace -- python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain-steps 6 --cpu-baseline-iters 0 > $O/r03_dense_bench_under_rocprof.json 2> $O/r03_dense_trace.log
# HBM traffic of the dominant kernel (PMC, separate passes)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/r03_fetch -- python3 scripts/bench_spmv.py config3 tiled 10 > $O/r03_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/r03_write -- python3 scripts/bench_spmv.py config3 tiled 10 > $O/r03_write.log 2>&1
# products on their own
{ python3 scripts/bench_spmv.py config3 tiled 200; python3 scripts/bench_spmv.py config2 tiled 200; } 2>&1 | grep -E "tiled geometry|avg|max abs err" > $O/r03_spmv.txt
{ for k in 2 4; do python3 scripts/bench_batch_products.py config3 $k 20; done; } 2>&1 | grep avg > $O/r03_batch_products.txt
{ for k in 4 8 16 32; do python3 scripts/bench_dense_batch.py 200000 8000 $k 5; done; } 2>&1 | grep avg > $O/r03_dense_batch_new.txt   # prepended to profiles/r03_dense_batch.txt by hand (its history stays)
python3 bench.py --config config4 --dense-storage float64 --steps 10 --warmup 3 --multi-chain-steps 6 --cpu-baseline-iters 0 > $O/r03_bench_config4_f64.json 2>> $O/r03_bench.err
python3 scripts/bench_small_batches.py 2>&1 | grep -E "^dense|^sparse" > $O/r03_small_batches.txt
BENCH_MIXED_PAIR=1 python3 scripts/bench_mixed.py 5 50 2>&1 | grep -E "pair products" > $O/r03_mixed_pair.txt
bash scripts/dense_batch_traffic.sh > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/bt_trace -- python3 scripts/batch_timeline.py config3 2 360 > $O/bt.log 2>&1
python3 scripts/batch_timeline.py --analyse $O/bt_trace > $O/r03_batch_timeline.txt; rm -rf $O/bt_trace
{ for c in 0 5 20 50; do python3 scripts/bench_mixed.py $c 100; done; python3 scripts/bench_mixed.py 20 100 0.02; } 2>&1 | grep -E "^all-binary|^mixed" > $O/r03_mixed.txt
{ for a in "1 1" "2 1" "1 2" "2 2"; do python3 scripts/overlap_probe.py $a 40; done; } 2>&1 | grep procs > $O/r03_overlap_probe.txt
(cd scripts/probes && hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_rate mfma_f64_rate.hip && /tmp/mfma_f64_rate) > $O/r03_mfma_f64_rate.txt 2>&1
# keep what is condensed, drop the bulky raw traces
python3 - <<'PY'
import glob, os, shutil
import pandas as pd
O = "gpurun_out"
for sub, out in (("r03_trace", "r03_kernel_stats.csv"), ("r03_dense_trace", "r03_dense_kernel_stats.csv")):
    f = sorted(glob.glob("%s/%s/*/*kernel_stats.csv" % (O, sub)))
    if f:
        d = pd.read_csv(f[-1])
        d["Name"] = d["Name"].map(lambda s: s.split("(")[0].replace("void ", "")[:70])
        d[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"]].head(40).to_csv("%s/%s" % (O, out), index=False)
rows = {}
for sub, ctr in (("r03_fetch", "FETCH_SIZE"), ("r03_write", "WRITE_SIZE")):
    f = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (O, sub)))
    if f:
        d = pd.read_csv(f[-1])
        d = d[(d.Counter_Name == ctr) & d.Kernel_Name.str.contains("tiled_spmv")]
        for g, grp in d.groupby("Grid_Size"):
            rows.setdefault("grid=%d" % (g // 1024), {})[ctr] = float(grp.Counter_Value.mean())
import json
for k, v in rows.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["total_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
json.dump({"hbm_traffic": rows, "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of scripts/bench_spmv.py config3 tiled 10; per launch; bytes = (2 FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction)"}, open("%s/r03_spmv_traffic.json" % O, "w"), indent=1)
for sub in ("r03_trace", "r03_dense_trace", "r03_fetch", "r03_write"):
    shutil.rmtree("%s/%s" % (O, sub), ignore_errors=True)
PY
