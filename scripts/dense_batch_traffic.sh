#!/bin/bash
# HBM traffic of the batched dense products (PMC, separate passes), one gpurun call.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dk_fetch -- python3 scripts/bench_dense_batch.py 200000 8000 16 2 > $O/dk_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dk_write -- python3 scripts/bench_dense_batch.py 200000 8000 16 2 > $O/dk_write.log 2>&1
python3 - <<'PY'
import glob, json, shutil
import pandas as pd
O = "gpurun_out"
rows = {}
for sub, ctr in (("dk_fetch", "FETCH_SIZE"), ("dk_write", "WRITE_SIZE")):
    f = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (O, sub)))
    if f:
        d = pd.read_csv(f[-1])
        d = d[d.Counter_Name == ctr]
        for name in ("dense_dot_kd_kernel", "dense_tdot_kd_kernel", "dense_fused_ring_kernel"):
            g = d[d.Kernel_Name.str.contains(name)]
            if len(g):
                rows.setdefault(name, {})[ctr] = float(g.Counter_Value.mean())
                rows[name]["launches"] = int(len(g))
for v in rows.values():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["total_bytes"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
json.dump({"hbm_traffic": rows,
           "algorithmic_bytes": {"matrix_200000x8008_f32": 200000 * 8008 * 4},
           "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of "
                  "scripts/bench_dense_batch.py 200000 8000 16 2; mean per launch; "
                  "bytes = (2 FETCH_SIZE + WRITE_SIZE) KB (gfx950 correction)"},
          open("%s/r03_dense_batch_traffic.json" % O, "w"), indent=1)
for sub in ("dk_fetch", "dk_write"):
    shutil.rmtree("%s/%s" % (O, sub), ignore_errors=True)
PY
cat $O/r03_dense_batch_traffic.json
