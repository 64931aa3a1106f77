"""Condenses rocprofv3 output under gpurun_out/<tag>_{trace,fetch,write,tcc}
into the small, tracked files under profiles/ that DESIGN.md and bench.py
cite.  Usage: python scripts/summarize_profiles.py r01"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    files = glob.glob(os.path.join(src, pattern))
    return pd.read_csv(files[0]) if files else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


stats = one("%s_trace/*/*kernel_stats.csv" % tag)
trace = one("%s_trace/*/*kernel_trace.csv" % tag)
summary = {}
if stats is not None:
    stats["Name"] = stats["Name"].map(short)
    keep = stats[["Name", "Calls", "TotalDurationNs", "AverageNs",
                  "Percentage", "MinNs", "MaxNs"]].head(30)
    keep.to_csv(os.path.join(dst, "%s_kernel_stats.csv" % tag), index=False)
if trace is not None:
    t = trace[trace.Kernel_Name.str.contains("tiled_spmv")].copy()
    t["dur_us"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
    grids = sorted(t.Grid_Size_X.unique())
    for g in grids:
        sel = t[t.Grid_Size_X == g]
        summary["tiled_spmv grid=%d" % (g // 1024)] = dict(
            launches=int(len(sel)), avg_us=float(sel.dur_us.mean()),
            median_us=float(sel.dur_us.median()), min_us=float(sel.dur_us.min()),
            lds_bytes=int(sel.LDS_Block_Size.iloc[0]),
            vgpr=int(sel.VGPR_Count.iloc[0]), sgpr=int(sel.SGPR_Count.iloc[0]))
pmc = {}
for key, pat in (("FETCH_SIZE", "%s_fetch"), ("WRITE_SIZE", "%s_write"),
                 ("TCC", "%s_tcc")):
    d = one((pat % tag) + "/*/*counter_collection.csv")
    if d is None:
        continue
    d = d[d.Kernel_Name.str.contains("tiled_spmv")]
    for (grid, cname), grp in d.groupby(["Grid_Size", "Counter_Name"]):
        pmc.setdefault("grid=%d" % (grid // 1024), {})[cname] = dict(
            mean=float(grp.Counter_Value.mean()),
            min=float(grp.Counter_Value.min()),
            max=float(grp.Counter_Value.max()), dispatches=int(len(grp)))
# HBM traffic per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes:
# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes
# of a 16-B/lane streaming read, so it is doubled; WRITE_SIZE is exact.
traffic = {}
for grid, c in pmc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[grid] = dict(
            read_bytes=2 * 1024 * c["FETCH_SIZE"]["mean"],
            write_bytes=1024 * c["WRITE_SIZE"]["mean"],
            total_bytes=2 * 1024 * c["FETCH_SIZE"]["mean"]
            + 1024 * c["WRITE_SIZE"]["mean"])
        if "TCC_HIT_sum" in c:
            traffic[grid]["l2_hit_rate"] = c["TCC_HIT_sum"]["mean"] / (
                c["TCC_HIT_sum"]["mean"] + c["TCC_MISS_sum"]["mean"])
out = dict(tag=tag, kernel_trace=summary, pmc=pmc, hbm_traffic=traffic)
with open(os.path.join(dst, "%s_spmv_profile.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
