"""Sums the FETCH_SIZE / WRITE_SIZE counters of two rocprofv3 --pmc passes of
scripts/iteration_traffic.py over the last `iters` Gibbs iterations and writes
profiles/r03_iteration_traffic.json (HBM bytes = 2 * FETCH_SIZE * 1024 +
WRITE_SIZE * 1024: on gfx950 FETCH_SIZE reports half the bytes of a 16-byte-
per-lane streaming read, MI355X_MICROARCH.md "HBM"; both counters are in KB).
Usage: python scripts/summarize_iteration_traffic.py <fetch dir> <write dir> <info.json> [out]"""
import glob
import json
import os
import sys

import pandas as pd

fetch_dir, write_dir, info_path = sys.argv[1:4]
out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(
    os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles",
    "r03_iteration_traffic.json")
info = json.load(open(info_path))
iters = info["iters"]


def window(directory, counter):
    f = sorted(glob.glob(os.path.join(directory, "*", "*counter_collection.csv")),
               key=os.path.getmtime)[-1]
    d = pd.read_csv(f)
    d = d[d.Counter_Name == counter].sort_values("Dispatch_Id")
    starts = d.index[d.Kernel_Name.str.contains("chain_prior_kernel")]
    first = starts[-iters]
    w = d.loc[first:]
    by = w.groupby(w.Kernel_Name.map(
        lambda s: s.split("(")[0].replace("void ", "").split("<")[0][-40:])
    ).Counter_Value.sum().sort_values(ascending=False)
    return float(w.Counter_Value.sum()), {k: float(v) for k, v in by.head(8).items()}


fetch_kb, fetch_by = window(fetch_dir, "FETCH_SIZE")
write_kb, write_by = window(write_dir, "WRITE_SIZE")
total = (2. * fetch_kb + write_kb) * 1024.
per_iter = total / iters
info.update(
    fetch_size_kb=fetch_kb, write_size_kb=write_kb,
    hbm_bytes_per_iteration=per_iter,
    model_over_measured=info["model_bytes_per_iteration"] / per_iter,
    fetch_kb_by_kernel=fetch_by, write_kb_by_kernel=write_by,
    how="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of "
        "scripts/iteration_traffic.py; counters summed from the %d-th last "
        "chain_prior_kernel to the end; bytes = (2 FETCH_SIZE + WRITE_SIZE) KB"
        % iters)
json.dump(info, open(out, "w"), indent=1)
print(json.dumps({k: info[k] for k in (
    "n_cg_iter", "model_bytes_per_iteration", "hbm_bytes_per_iteration",
    "model_over_measured")}))
