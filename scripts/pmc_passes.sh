#!/bin/bash
# Collects wave-state / LDS / memory-pipe counters of the operator kernels in
# separate rocprofv3 --pmc passes (never combined with tracing).
# Usage (on the GPU box): bash scripts/pmc_passes.sh <tag>
export TMPDIR=/tmp
tag=${1:-pmc}
i=0
while read -r ctrs; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/${tag}_p$i -- python3 scripts/bench_spmv.py config3 tiled 10 > gpurun_out/${tag}_p$i.log 2>&1
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_BUSY_CU_CYCLES
LIST
python3 - "$tag" <<'PY'
import glob, sys
import pandas as pd
tag = sys.argv[1]
rows = []
for f in sorted(glob.glob("gpurun_out/%s_p*/*/*counter_collection.csv" % tag)):
    d = pd.read_csv(f)
    d = d[d.Kernel_Name.str.contains("tiled_spmv")]
    for (g, c), grp in d.groupby(["Grid_Size", "Counter_Name"]):
        rows.append((g // 1024, c, grp.Counter_Value.mean()))
out = pd.DataFrame(rows, columns=["grid", "counter", "mean"]).pivot(index="counter", columns="grid", values="mean")
print(out.to_string())
out.to_csv("gpurun_out/%s_summary.csv" % tag)
PY
