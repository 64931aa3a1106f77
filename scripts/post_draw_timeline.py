"""Kernel-trace timeline of the stretch between two CG solves of a device chain.
Run on the GPU box (rocprofv3 writes the trace, this script condenses it):

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- \
        python3 bench.py --multi-chain 0 --cpu-baseline-iters 0 --repeat 1
    python3 scripts/post_draw_timeline.py gpurun_out/tl [n_th finish kernel]

Prints, for one mid-run iteration, every kernel from cg_finish_kernel of one
solve to the first full-length product of the next: queue, start, end, duration
(microseconds from the start of cg_finish_kernel).  Under the profiler the host
launches slower than in a plain run: gaps are upper bounds."""
import glob
import os
import sys

import pandas as pd

src = sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 200
files = sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"),
                         recursive=True), key=os.path.getmtime)
tr = pd.read_csv(files[-1]).sort_values("Start_Timestamp").reset_index(drop=True)
name = tr["Kernel_Name"].astype(str)
fin = tr.index[name.str.contains("cg_finish_kernel")]
i0 = fin[min(nth, len(fin) - 2)]
i1 = fin[min(nth, len(fin) - 2) + 1]
t0 = tr.loc[i0, "Start_Timestamp"]
rows = tr.loc[i0:i1]
queues = {q: k for k, q in enumerate(sorted(rows["Queue_Id"].unique()))}
print("%-46s %5s %9s %9s %8s" % ("name", "queue", "start_us", "end_us", "dur_us"))
shown = 0
for _, r in rows.iterrows():
    nm = str(r["Kernel_Name"]).replace("void ", "").replace("bbx::", "")
    nm = nm.split("(")[0][:46]
    s, e = (r["Start_Timestamp"] - t0) / 1e3, (r["End_Timestamp"] - t0) / 1e3
    # the CG loop itself: only its first two products
    if "tiled_spmv" in nm or "cg_direction" in nm or "tdot_finalize" in nm:
        shown += 1
        if shown > 6:
            continue
    print("%-46s %5d %9.1f %9.1f %8.1f" % (nm, queues[r["Queue_Id"]], s, e, e - s))
