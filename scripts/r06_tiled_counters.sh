#!/bin/bash
# Round 6: hardware counters of tiled_spmv_kernel at config 3 (1M x 50k), one
# rocprofv3 --pmc pass per line (never combined with tracing; the program
# itself after `--`), condensed into gpurun_out/r06_tiled_counters.json.
#   bash scripts/r06_tiled_counters.sh [reps]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
reps=${1:-10}
i=0
while read -r ctrs; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $O/r06_pmc_p$i -- python3 scripts/bench_spmv.py config3 tiled $reps > $O/r06_pmc_p$i.log 2>&1 || echo "pass $i ($ctrs) failed: $(tail -2 $O/r06_pmc_p$i.log)"
done <<'LIST'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_ACTIVE_INST_ANY
SQ_BUSY_CU_CYCLES SQ_CYCLES GRBM_GUI_ACTIVE SQ_WAVES
FETCH_SIZE
WRITE_SIZE
LIST
python3 - <<'PY'
import glob, json
import pandas as pd
out = {}
for f in sorted(glob.glob("gpurun_out/r06_pmc_p*/*/*counter_collection.csv")):
    d = pd.read_csv(f)
    d = d[d.Kernel_Name.str.contains("tiled_spmv")]
    for (g, c), grp in d.groupby(["Grid_Size", "Counter_Name"]):
        out.setdefault("grid=%d" % (g // 1024), {})[c] = float(grp.Counter_Value.mean())
for g, c in out.items():
    def r(a, b):
        return round(c[a] / c[b], 4) if a in c and b in c and c[b] else None
    c["derived"] = {
        "lds_bank_conflict_share_of_lds_cycles": r("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
        "wait_inst_any_share_of_wave_cycles": r("SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
        "wait_inst_lds_share_of_wave_cycles": r("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
        "vmem_issue_share_of_busy_cycles": r("SQ_INST_CYCLES_VMEM", "SQ_BUSY_CYCLES"),
        "active_inst_lds_share_of_busy_cycles": r("SQ_ACTIVE_INST_LDS", "SQ_BUSY_CYCLES"),
        "active_inst_valu_share_of_busy_cycles": r("SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES"),
        "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c else None,
        "tcp_pending_stall_per_read_req": r("TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum"),
        "tcp_read_latency_cycles": r("TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum"),
        "valu_per_vmem_read": r("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD"),
        "lds_per_vmem_read": r("SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"),
        "mean_waves_in_flight": r("SQ_LEVEL_WAVES", "SQ_BUSY_CYCLES"),
        "mean_vmem_in_flight_per_cu_cycle": r("SQ_INST_LEVEL_VMEM", "SQ_BUSY_CU_CYCLES"),
        "hbm_bytes_per_launch": (2048. * c["FETCH_SIZE"] + 1024. * c.get("WRITE_SIZE", 0.))
        if "FETCH_SIZE" in c else None,
    }
json.dump({"what": "rocprofv3 --pmc, one pass per counter group, mean per launch of "
                   "tiled_spmv_kernel by launch grid (workgroups); scripts/bench_spmv.py "
                   "config3 tiled; FETCH_SIZE doubled (gfx950 correction)",
           "by_grid": out}, open("gpurun_out/r06_tiled_counters.json", "w"), indent=1)
print(json.dumps({g: c["derived"] for g, c in out.items()}, indent=1))
PY
for d in $O/r06_pmc_p*/; do rm -rf $d; done
