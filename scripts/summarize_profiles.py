"""Condenses rocprofv3 output (written on the GPU box by scripts/profile_r04.sh)
into the small, tracked files under profiles/ that DESIGN.md and bench.py cite.

    python scripts/summarize_profiles.py r04 [src_dir] [dst_dir]

Writes <tag>_spmv_profile.json and <tag>_dense_profile.json:
  kernel_trace   per (kernel instantiation, launch grid): launches, launches
                 that returned at entry (the CG solve they belonged to had
                 stopped: a few microseconds, not executions of the kernel),
                 average / median / minimum duration of the others
  pmc            raw counter means per (kernel, grid): FETCH_SIZE, WRITE_SIZE
                 (KiB), dispatches
  hbm_traffic    bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (gfx950: the
                 fetch counter reports half the bytes of a 16-byte-per-lane
                 stream; MI355X_MICROARCH.md "HBM"), and -- where the bench line
                 names the kernel's algorithmic bytes -- their ratio
  dominant       the product kernels of the CG loop and of the K = 2 batch with
                 duration, algorithmic bytes and fraction of the 8 TB/s peak
plus <tag>_kernel_stats.csv / <tag>_dense*_kernel_stats.csv and the bench lines.
"""
import glob
import json
import os
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out")
dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
PEAK = 8000.0   # GB/s


def one(pattern):
    files = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)
    return pd.read_csv(files[-1]) if files else None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("bbx::", "")[:72]


def kernel_stats(sub, out_name):
    stats = one("%s_%s/*/*kernel_stats.csv" % (tag, sub))
    if stats is None:
        return
    stats["Name"] = stats["Name"].map(short)
    stats[["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
           "MinNs", "MaxNs"]].head(40).to_csv(os.path.join(dst, out_name),
                                              index=False)


def trace_summary(sub, pattern):
    trace = one("%s_%s/*/*kernel_trace.csv" % (tag, sub))
    out = {}
    if trace is None:
        return out
    t = trace[trace.Kernel_Name.str.contains(pattern)].copy()
    t["dur_us"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
    t["short"] = t.Kernel_Name.map(short)
    for (name, g, wg), sel in t.groupby(["short", "Grid_Size_X",
                                         "Workgroup_Size_X"]):
        # launches enqueued past the stopping iteration of their CG solve see
        # the stop flag and return at entry: 2-4 us against tens of us
        floor_us = .5 * sel.dur_us.median()
        n_all = len(sel)
        sel = sel[sel.dur_us >= floor_us]
        out["%s grid=%d" % (name, g // wg)] = dict(
            launches=int(len(sel)), returned_at_entry=int(n_all - len(sel)),
            avg_us=round(float(sel.dur_us.mean()), 3),
            median_us=round(float(sel.dur_us.median()), 3),
            min_us=round(float(sel.dur_us.min()), 3),
            vgpr=int(sel.VGPR_Count.iloc[0]), sgpr=int(sel.SGPR_Count.iloc[0]))
    return out


def pmc(sub, pattern):
    d = one("%s_%s/*/*counter_collection.csv" % (tag, sub))
    out = {}
    if d is None:
        return out
    d = d[d.Kernel_Name.str.contains(pattern)].copy()
    d["short"] = d.Kernel_Name.map(short)
    for (name, grid, wg, cname), grp in d.groupby(
            ["short", "Grid_Size", "Workgroup_Size", "Counter_Name"]):
        # (dispatches that returned at entry move no bytes)
        grp = grp[grp.Counter_Value >= .5 * grp.Counter_Value.median()]
        out.setdefault("%s grid=%d" % (name, grid // wg), {})[cname] = dict(
            mean=float(grp.Counter_Value.mean()),
            min=float(grp.Counter_Value.min()),
            max=float(grp.Counter_Value.max()), dispatches=int(len(grp)))
    return out


def merge(*dicts):
    out = {}
    for d in dicts:
        for key, c in d.items():
            out.setdefault(key, {}).update(c)
    return out


def traffic_of(counters):
    out = {}
    for key, c in counters.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rd = 2 * 1024 * c["FETCH_SIZE"]["mean"]
            wr = 1024 * c["WRITE_SIZE"]["mean"]
            out[key] = dict(read_bytes=rd, write_bytes=wr, total_bytes=rd + wr)
    return out


def bench_line(name):
    path = os.path.join(src, "%s_%s.json" % (tag, name))
    try:
        with open(path) as fh:
            return json.loads(fh.read().strip().splitlines()[-1])
    except Exception:      # noqa: BLE001
        return None


def find(table, must, grid):
    for key, val in table.items():
        if all(m in key for m in must) and key.endswith("grid=%d" % grid):
            return key, val
    return None, None


# ---------------------------------------------------------------- config 3
kernel_stats("trace", "%s_kernel_stats.csv" % tag)
trace = trace_summary("trace", "tiled_spmv|tdot_finalize|cg_|b_finalize|b_dir")
counters = merge(pmc("loop_fetch", "tiled_spmv|tdot_finalize"),
                 pmc("loop_write", "tiled_spmv|tdot_finalize"),
                 pmc("k2_fetch", "tiled_spmv"), pmc("k2_write", "tiled_spmv"))
traffic = traffic_of(counters)
line = bench_line("bench_under_rocprof") or bench_line("bench")
plain = bench_line("bench")
dominant = {}
if line and line["config"].get("launch_grids"):
    grids = line["config"]["launch_grids"]
    other = (plain or line)["roofline"]["other"]
    fold = line["config"].get("cg_launches_per_iteration") == 3
    want = {
        "dot (X~ v inside the CG loop%s)" % (", direction step folded in"
                                             if fold else ""):
            # (VALS, WIDE, KP, FOLD, DENSEP, PACK: either id format)
            (["tiled_spmv_kernel<false, true, 0, %s, false" % ("true" if fold
                                                                else "false")],
             grids["X"], other["dot"]["bytes"]),
        "tdot (X~^T w main kernel)":
            (["tiled_spmv_kernel<false, true, 0, false, false"], grids["Xt"],
             other["tdot"]["bytes"]),
    }
    mc = (plain or line).get("multi_chain") or {}
    k2 = mc.get("k=2") or {}
    if k2.get("launch_grids"):
        want["dot, K = 2 batch"] = (["tiled_spmv_kernel<false, true, 1, false"],
                                    k2["launch_grids"]["X"], k2["dot"]["bytes"])
        want["tdot, K = 2 batch"] = (["tiled_spmv_kernel<false, true, 1, false"],
                                     k2["launch_grids"]["Xt"], k2["tdot"]["bytes"])
    for label, (must, grid, nbytes) in want.items():
        key, tr = find(trace, must, grid)
        entry = dict(kernel=key, algorithmic_bytes=int(nbytes))
        if tr:
            entry.update(trace_avg_us=tr["avg_us"], launches=tr["launches"],
                         returned_at_entry=tr["returned_at_entry"],
                         gbs=round(nbytes / tr["avg_us"] / 1e3, 1),
                         frac_of_8TBs=round(nbytes / tr["avg_us"] / 1e3 / PEAK, 4))
        _, tf = find(traffic, must, grid)
        if tf:
            entry.update(hbm_bytes_pmc=int(tf["total_bytes"]),
                         traffic_over_algorithmic=round(tf["total_bytes"] / nbytes, 4))
        dominant[label] = entry
out = dict(tag=tag, kernel_trace=trace, pmc=counters, hbm_traffic=traffic,
           dominant=dominant,
           how="kernel_trace: rocprofv3 --kernel-trace of `python3 bench.py "
               "--cpu-baseline-iters 0` (timed region, burn-in and the "
               "multi_chain blocks; the K = 2 kernels are its k=2 block); pmc: "
               "separate --pmc FETCH_SIZE / WRITE_SIZE passes of "
               "scripts/iteration_traffic.py 5 60 (the CG loop's kernels inside "
               "a chain) and scripts/bench_batch_products.py config3 2 10"
               "; the X~ v kernel inside the CG loop also reads the row scale "
               "Omega (8 n = 8.0 MB at n = 1e6), which `algorithmic_bytes` "
               "(format bytes + vector in + vector out, bbx_design_timed_bytes) "
               "does not count; "
               "bench.iteration_bytes credits Omega separately")
# bench.py's `traffic` lookup (committed_traffic) reads hbm_traffic["grid=N"]
for marker in ("tiled_spmv_kernel<false, true, 0, false, false",
               "tiled_spmv_kernel<false, true, 0, true, false"):
    for key, val in list(traffic.items()):
        if marker in key:
            out["hbm_traffic"].setdefault(key.split(" ")[-1], val)
with open(os.path.join(dst, "%s_spmv_profile.json" % tag), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)

# whole-iteration traffic with the folded direction step (pins bench.iteration_bytes)
run = os.path.join(src, "%s_iteration_run.json" % tag)
if os.path.exists(run):
    info = json.load(open(run))
    tot = {}
    for sub, ctr in (("loop_fetch", "FETCH_SIZE"), ("loop_write", "WRITE_SIZE")):
        d = one("%s_%s/*/*counter_collection.csv" % (tag, sub))
        if d is None:
            continue
        d = d[d.Counter_Name == ctr].sort_values("Dispatch_Id")
        # the last `iters` Gibbs iterations start at the iters-th last prior kernel
        starts = d[d.Kernel_Name.str.contains("chain_prior_kernel")].Dispatch_Id
        if len(starts) >= info["iters"]:
            first = starts.iloc[-info["iters"]]
            tot[ctr] = float(d[d.Dispatch_Id >= first].Counter_Value.sum())
    if len(tot) == 2:
        info.update(fetch_size_kb=tot["FETCH_SIZE"], write_size_kb=tot["WRITE_SIZE"],
                    hbm_bytes_per_iteration=(2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"])
                    * 1024. / info["iters"])
        info["model_over_measured"] = info["model_bytes_per_iteration"] / \
            info["hbm_bytes_per_iteration"]
        with open(os.path.join(dst, "%s_iteration_traffic.json" % tag), "w") as fh:
            json.dump(info, fh, indent=1, sort_keys=True)

# ---------------------------------------------------------------- config 4
dense = dict(tag=tag)
for label, sub, bench_name in (("f32", "dense", "bench_config4"),
                               ("f64", "dense64", "bench_config4_f64")):
    kernel_stats(sub + "_trace", "%s_%s_kernel_stats.csv" % (tag, sub))
    tr = trace_summary(sub + "_trace", "dense_|tdot_finalize")
    c = merge(pmc(sub + "_fetch", "dense_fused"), pmc(sub + "_write", "dense_fused"))
    tf = traffic_of(c)
    entry = dict(kernel_trace=tr, pmc=c, hbm_traffic=tf)
    bl = bench_line(bench_name)
    if bl:
        nbytes = bl["roofline"]["algorithmic_bytes_per_launch"]
        key = next((k for k in tr if "dense_fused" in k), None)
        if key:
            entry["dominant"] = dict(
                kernel=key, algorithmic_bytes=int(nbytes),
                trace_avg_us=tr[key]["avg_us"], launches=tr[key]["launches"],
                returned_at_entry=tr[key]["returned_at_entry"],
                frac_of_8TBs=round(nbytes / tr[key]["avg_us"] / 1e3 / PEAK, 4))
            tkey = next((k for k in tf if "dense_fused" in k), None)
            if tkey:
                entry["dominant"].update(
                    hbm_bytes_pmc=int(tf[tkey]["total_bytes"]),
                    traffic_over_algorithmic=round(
                        tf[tkey]["total_bytes"] / nbytes, 4))
    dense[label] = entry
with open(os.path.join(dst, "%s_dense_profile.json" % tag), "w") as fh:
    json.dump(dense, fh, indent=1, sort_keys=True)

# ---------------------------------------------------------------- bench lines
for path in glob.glob(os.path.join(src, "%s_bench*.json" % tag)) + \
        glob.glob(os.path.join(src, "%s_dense*_bench_under_rocprof.json" % tag)):
    if os.path.getsize(path) > 0:
        shutil.copy(path, os.path.join(dst, os.path.basename(path)))
print(json.dumps(dominant, indent=1))
print(json.dumps({k: v.get("dominant") for k, v in dense.items()
                  if isinstance(v, dict)}, indent=1))
