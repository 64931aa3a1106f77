#!/bin/bash
# HBM traffic of the K = 2 tiled products (PMC, separate passes), one gpurun call.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bk_fetch -- python3 scripts/bench_batch_products.py ${CFG:-config3} ${K:-2} 10 > $O/bk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bk_write -- python3 scripts/bench_batch_products.py ${CFG:-config3} ${K:-2} 10 > $O/bk_write.log 2>&1
python3 - <<'PY'
import glob, json, os, shutil
import pandas as pd
O = "gpurun_out"
rows = {}
for sub, ctr in (("bk_fetch", "FETCH_SIZE"), ("bk_write", "WRITE_SIZE")):
    f = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (O, sub)))
    if f:
        d = pd.read_csv(f[-1])
        d = d[(d.Counter_Name == ctr) & d.Kernel_Name.str.contains("tiled_spmv_kernel<false, true, %d>" % (int(os.environ.get("K", "2")) // 2), regex=False)]
        for g, grp in d.groupby("Grid_Size"):
            rows.setdefault("grid=%d" % (g // 1024), {})[ctr] = float(grp.Counter_Value.mean())
            rows["grid=%d" % (g // 1024)]["launches"] = int(len(grp))
for v in rows.values():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["total_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
log = open("%s/bk_fetch.log" % O).read()
alg = [l for l in log.splitlines() if l.startswith("K=")]
json.dump({"hbm_traffic": rows, "bench_lines": alg,
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of "
                  "scripts/bench_batch_products.py config3 2 10; tiled_spmv_kernel<false, true, 1> "
                  "(grid 253 = X V, grid 255 = X^T W); mean per launch; bytes = "
                  "(2 FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction)"},
          open("%s/r03_batch_traffic_%s_k%s.json" % (O, os.environ.get("CFG", "config3"), os.environ.get("K", "2")), "w"), indent=1)
for sub in ("bk_fetch", "bk_write"):
    shutil.rmtree("%s/%s" % (O, sub), ignore_errors=True)
PY
cat $O/r03_batch_traffic_${CFG:-config3}_k${K:-2}.json
