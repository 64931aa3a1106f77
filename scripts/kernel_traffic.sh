#!/bin/bash
# Per-kernel HBM traffic of any script (two PMC passes): bash scripts/kernel_traffic.sh <out.txt> <python script + args>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out; out=$1; shift
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/kt_fetch -- python3 "$@" > $O/kt_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/kt_write -- python3 "$@" > $O/kt_write.log 2>&1
python3 - $out <<'PY'
import glob, shutil, sys
import pandas as pd
O = "gpurun_out"
tab = {}
for sub, ctr in (("kt_fetch", "FETCH_SIZE"), ("kt_write", "WRITE_SIZE")):
    f = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (O, sub)))[-1]
    d = pd.read_csv(f)
    d = d[d.Counter_Name == ctr]
    name = d.Kernel_Name.str.replace("void ", "").str.split("(").str[0].str[:60]
    g = d.assign(k=name).groupby("k").Counter_Value.agg(["count", "mean"])
    for k, r in g.iterrows():
        tab.setdefault(k, {})[ctr] = (int(r["count"]), float(r["mean"]))
rows = []
for k, v in tab.items():
    n = v.get("FETCH_SIZE", (0, 0))[0]
    fe = v.get("FETCH_SIZE", (0, 0.))[1]
    wr = v.get("WRITE_SIZE", (0, 0.))[1]
    rows.append((n * (2 * fe + wr) * 1024, k, n, 2 * fe * 1024, wr * 1024))
rows.sort(reverse=True)
with open(sys.argv[1], "w") as fh:
    fh.write("# per launch: bytes read (2 x FETCH_SIZE KB) and written (WRITE_SIZE KB), mean over the launches\n")
    for tot, k, n, fe, wr in rows[:30]:
        fh.write("%-62s launches %5d  read %10.3f MB  written %9.3f MB  (total %8.1f MB)\n"
                 % (k, n, fe / 1e6, wr / 1e6, tot / 1e6))
for sub in ("kt_fetch", "kt_write"):
    shutil.rmtree("%s/%s" % (O, sub), ignore_errors=True)
PY
cat $out
