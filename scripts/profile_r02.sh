#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun).  Kernel traces and PMC
# counters are ALWAYS separate rocprofv3 runs.
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out
# --- config 3 (headline): kernel trace of the bench run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_trace -- python3 bench.py --cpu-baseline-iters 0 > $O/r02_bench_under_rocprof.json 2> $O/r02_trace.err
# --- config 3: HBM traffic + L2 + LDS conflict counters of the operator kernels
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "tcc:TCC_HIT_sum TCC_MISS_sum" "lds:SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "wait:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/r02_$name -- python3 scripts/bench_spmv.py config3 tiled 10 > $O/r02_$name.log 2>&1
done
# the same LDS counters with the ascending-id entry order (A/B of the bank-aware order)
BBX_TILED_BANKS=0 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d $O/r02_lds_plain -- python3 scripts/bench_spmv.py config3 tiled 10 > $O/r02_lds_plain.log 2>&1
# --- config 4 (dense f32): kernel trace of the bench run + traffic of the fused kernel
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r02_dense_trace -- python3 bench.py --config config4 --steps 6 --warmup 2 --burnin 4 --cpu-baseline-iters 0 > $O/r02_dense_bench_under_rocprof.json 2> $O/r02_dense_trace.err
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --output-format csv -d $O/r02_dense_$name -- python3 bench.py --config config4 --steps 2 --warmup 1 --burnin 1 --cpu-baseline-iters 0 > $O/r02_dense_$name.log 2>&1
done
bash scripts/pmc_dense.sh r02_dense > $O/r02_dense_pmc.txt 2>&1
# --- plain bench lines (no profiler attached)
python3 bench.py > $O/r02_bench.json 2> $O/r02_bench.err
python3 bench.py --config config4 --steps 10 --warmup 2 --burnin 5 > $O/r02_bench_config4.json 2> $O/r02_bench_config4.err
python3 bench.py --config config2 --cpu-baseline-iters 0 > $O/r02_bench_config2.json 2> $O/r02_bench_config2.err
ls $O | grep r02_ | head -50
