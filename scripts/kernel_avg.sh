#!/bin/bash
# average duration of the library's kernels in a short traced bench run
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/tmp_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tmp_trace -- python3 bench.py --cpu-baseline-iters 0 --steps 30 --burnin 100 "$@" > /dev/null 2>&1
python3 - <<'PY'
import glob
import pandas as pd
s = pd.read_csv(sorted(glob.glob("gpurun_out/tmp_trace/*/*kernel_stats.csv"))[-1])
s = s[s.Name.str.contains("bbx::")]
s["Name"] = s.Name.str.replace("void ", "").str.slice(0, 44)
print(s[["Name", "Calls", "AverageNs", "Percentage"]].head(9).to_string(index=False))
PY
