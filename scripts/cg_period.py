"""CG iteration period from a rocprofv3 kernel trace: the median start-to-start
interval of consecutive X~ v launches of the CG loop (intervals beyond 3x the
median belong to solve boundaries and are dropped), and the busy / gap split.
Usage: python scripts/cg_period.py <trace dir> [grid of the X~ v kernel]"""
import glob
import sys

import pandas as pd

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
t = pd.read_csv(f).sort_values("Start_Timestamp")
t["name"] = t.Kernel_Name.map(lambda s: s.split("(")[0].replace("void ", "").replace("bbx::", "")[:60])
t["grid"] = t.Grid_Size_X // t.Workgroup_Size_X
t["dur"] = (t.End_Timestamp - t.Start_Timestamp) / 1e3
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 253
dot = t[t.name.str.contains("tiled_spmv_kernel<false, true, 0") & (t.grid == grid) & (t.dur > 20)]
starts = dot.Start_Timestamp.values
iv = (starts[1:] - starts[:-1]) / 1e3
med = float(pd.Series(iv).median())
keep = iv[iv < 3 * med]
print("X~ v launches %d; start-to-start interval: median %.2f us, mean of in-solve intervals %.2f us (n = %d)"
      % (len(dot), med, keep.mean(), len(keep)))
# busy time inside one period: sum of kernel durations between two consecutive dot starts
tt = t[(t.Start_Timestamp >= starts[len(starts) // 2]) & (t.Start_Timestamp < starts[len(starts) // 2 + 200])]
busy = tt.dur.sum() / 200.
print("mean busy time per period over 200 periods mid-run: %.2f us; kernels per period %.2f" % (busy, len(tt) / 200.))
print(tt.groupby(["name", "grid"]).dur.agg(["count", "mean"]).sort_values("count", ascending=False).head(8).to_string())
