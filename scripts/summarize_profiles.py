"""Condenses rocprofv3 output under gpurun_out/<tag>_* (written on the GPU box
by scripts/profile_r02.sh) into the small, tracked files under profiles/ that
DESIGN.md and bench.py cite.  Usage: python scripts/summarize_profiles.py r02"""
import glob
import json
import os
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    # gpurun merges every call's output into the same directories: the newest
    # file of a pass is the one that belongs to the current code
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return pd.read_csv(files[-1]) if files else None


def short(name):
    return name.split("(")[0].replace("void ", "")[:70]


def kernel_stats(sub, out_name):
    stats = one("%s_%s/*/*kernel_stats.csv" % (tag, sub))
    if stats is None:
        return
    stats["Name"] = stats["Name"].map(short)
    keep = stats[["Name", "Calls", "TotalDurationNs", "AverageNs",
                  "Percentage", "MinNs", "MaxNs"]].head(30)
    keep.to_csv(os.path.join(dst, out_name), index=False)


def trace_summary(sub, pattern):
    """Per (kernel, grid) launch statistics from a kernel trace."""
    trace = one("%s_%s/*/*kernel_trace.csv" % (tag, sub))
    out = {}
    if trace is None:
        return out
    t = trace[trace.Kernel_Name.str.contains(pattern)].copy()
    t["dur_us"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
    t["short"] = t.Kernel_Name.map(short)
    for (name, g, wg), sel in t.groupby(["short", "Grid_Size_X",
                                         "Workgroup_Size_X"]):
        # launches that found their CG solve already stopped return at entry
        # (a few microseconds): they are not executions of the kernel
        floor_us = .5 * sel.dur_us.median()
        n_all = len(sel)
        sel = sel[sel.dur_us >= floor_us]
        out["%s grid=%d" % (name.split("::")[-1], g // wg)] = dict(
            launches=int(len(sel)), returned_at_entry=int(n_all - len(sel)),
            avg_us=float(sel.dur_us.mean()),
            median_us=float(sel.dur_us.median()),
            min_us=float(sel.dur_us.min()),
            vgpr=int(sel.VGPR_Count.iloc[0]), sgpr=int(sel.SGPR_Count.iloc[0]))
    return out


def pmc(sub, pattern, wg=1024):
    d = one("%s_%s/*/*counter_collection.csv" % (tag, sub))
    out = {}
    if d is None:
        return out
    d = d[d.Kernel_Name.str.contains(pattern)]
    for (grid, cname), grp in d.groupby(["Grid_Size", "Counter_Name"]):
        # (same filter: dispatches that returned at entry move no bytes)
        grp = grp[grp.Counter_Value >= .5 * grp.Counter_Value.median()]
        out.setdefault("grid=%d" % (grid // wg), {})[cname] = dict(
            mean=float(grp.Counter_Value.mean()),
            min=float(grp.Counter_Value.min()),
            max=float(grp.Counter_Value.max()), dispatches=int(len(grp)))
    return out


def merge(*dicts):
    out = {}
    for d in dicts:
        for grid, c in d.items():
            out.setdefault(grid, {}).update(c)
    return out


def traffic_of(counters):
    """HBM traffic per launch, corrected as MI355X_MICROARCH.md "HBM"
    prescribes: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports
    half the bytes of a 16-B/lane streaming read, so it is doubled; WRITE_SIZE
    is exact."""
    out = {}
    for grid, c in counters.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd = 2 * 1024 * c["FETCH_SIZE"]["mean"]
            wr = 1024 * c["WRITE_SIZE"]["mean"]
            out[grid] = dict(read_bytes=rd, write_bytes=wr,
                             total_bytes=rd + wr)
            if "TCC_HIT_sum" in c:
                out[grid]["l2_hit_rate"] = c["TCC_HIT_sum"]["mean"] / (
                    c["TCC_HIT_sum"]["mean"] + c["TCC_MISS_sum"]["mean"])
    return out


# ---- config 3: the tiled operator kernels
kernel_stats("trace", "%s_kernel_stats.csv" % tag)
spmv = merge(pmc("fetch", "tiled_spmv"), pmc("write", "tiled_spmv"),
             pmc("tcc", "tiled_spmv"), pmc("lds", "tiled_spmv"),
             pmc("wait", "tiled_spmv"))
lds_plain = pmc("lds_plain", "tiled_spmv")
lds_ab = {}
for grid in spmv:
    row = {}
    for label, src_c in (("bank_aware", spmv.get(grid, {})),
                         ("ascending_ids", lds_plain.get(grid, {}))):
        if "SQ_LDS_BANK_CONFLICT" in src_c:
            row[label] = dict(
                conflict_cycles=src_c["SQ_LDS_BANK_CONFLICT"]["mean"],
                lds_cycles=src_c["SQ_LDS_IDX_ACTIVE"]["mean"],
                conflict_share=src_c["SQ_LDS_BANK_CONFLICT"]["mean"]
                / src_c["SQ_LDS_IDX_ACTIVE"]["mean"])
    if row:
        lds_ab[grid] = row
out = dict(tag=tag,
           kernel_trace=trace_summary("trace", "tiled_spmv|tdot_finalize|cg_"),
           pmc=spmv, hbm_traffic=traffic_of(spmv), lds_bank_conflicts=lds_ab)
with open(os.path.join(dst, "%s_spmv_profile.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)

# ---- config 4: the dense kernels
kernel_stats("dense_trace", "%s_dense_kernel_stats.csv" % tag)
dense_c = merge(pmc("dense_fetch", "dense_fused", wg=1024),
                pmc("dense_write", "dense_fused", wg=1024))
dense = dict(tag=tag,
             kernel_trace=trace_summary("dense_trace", "dense_|tdot_finalize"),
             pmc=dense_c, hbm_traffic=traffic_of(dense_c))
csv = os.path.join(src, "%s_dense_pmc_summary.csv" % tag)
if os.path.exists(csv):
    tab = pd.read_csv(csv).set_index("counter")
    dense["gemv_valu_vs_mfma_counters"] = {
        k: {c: float(v) for c, v in row.items()}
        for k, row in tab.to_dict(orient="index").items()}
ab = os.path.join(src, "%s_dense_mfma_ab.txt" % tag)
if os.path.exists(ab):
    dense["gemv_valu_vs_mfma_timing"] = [
        ln.strip() for ln in open(ab) if ln.startswith("BBX_DENSE_MFMA")]
with open(os.path.join(dst, "%s_dense_profile.json" % tag), "w") as fh:
    json.dump(dense, fh, indent=1, sort_keys=True)

# ---- the bench lines of the profiled and the plain runs
for name in ("bench", "bench_config4", "bench_config2", "bench_under_rocprof",
             "dense_bench_under_rocprof"):
    path = os.path.join(src, "%s_%s.json" % (tag, name))
    if os.path.exists(path) and os.path.getsize(path) > 0:
        with open(path) as fh, open(os.path.join(
                dst, "%s_%s.json" % (tag, name)), "w") as out_fh:
            out_fh.write(fh.read())
for name in ("mall_policy", "inflight", "lds_atomic", "ab_2wg"):
    path = os.path.join(src, "%s_%s.txt" % (tag, name))
    if os.path.exists(path):
        with open(path) as fh, open(os.path.join(
                dst, "%s_%s.txt" % (tag, name)), "w") as out_fh:
            out_fh.write(fh.read())
print(json.dumps({k: out[k] for k in ("hbm_traffic", "lds_bank_conflicts")},
                 indent=1, sort_keys=True))
print(json.dumps({k: dense[k] for k in ("hbm_traffic",) if k in dense},
                 indent=1))
