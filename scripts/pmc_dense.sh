#!/bin/bash
# Instruction-mix / busy counters of the dense GEMV variants (separate --pmc
# passes, never combined with tracing).  Usage: bash scripts/pmc_dense.sh <tag>
export TMPDIR=/tmp
tag=${1:-r02_dense}
for mf in 0 1; do
  i=0
  while read -r ctrs; do
    i=$((i+1))
    BBX_DENSE_MFMA=$mf timeout 300 rocprofv3 --pmc $ctrs --output-format csv \
      -d gpurun_out/${tag}_mfma${mf}_p$i -- python3 scripts/ab_dense_mfma.py 200000 8000 5 \
      > gpurun_out/${tag}_mfma${mf}_p$i.log 2>&1
  done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM
SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F64
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CU_CYCLES
LIST
done
python3 - "$tag" <<'PY'
import glob, sys
import pandas as pd
tag = sys.argv[1]
rows = []
for mf in (0, 1):
    for f in sorted(glob.glob("gpurun_out/%s_mfma%d_p*/*/*counter_collection.csv" % (tag, mf))):
        d = pd.read_csv(f)
        d = d[d.Kernel_Name.str.contains("dense_dot")]
        for c, grp in d.groupby("Counter_Name"):
            rows.append(("mfma" if mf else "valu", c, grp.Counter_Value.mean()))
out = pd.DataFrame(rows, columns=["variant", "counter", "mean"]).pivot_table(
    index="counter", columns="variant", values="mean")
print(out.to_string())
out.to_csv("gpurun_out/%s_pmc_summary.csv" % tag)
PY
