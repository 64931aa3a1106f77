#!/bin/bash
# Round-4 profiles, one gpurun call (kernel traces and --pmc passes are always
# separate rocprofv3 runs).  Raw output under gpurun_out/r04_*; condensed into
# gpurun_out/r04p/ by `scripts/summarize_profiles.py r04 gpurun_out gpurun_out/r04p`
# (run at the end of this script, on the box, so that the bulky traces need not
# travel), and copied from there into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/r04p
# ---- plain bench lines
python3 bench.py > $O/r04_bench.json 2> $O/r04_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r04_bench_driverflags.json 2>> $O/r04_bench.err
python3 bench.py --cg-fold 1 --multi-chain 0 --cpu-baseline-iters 0 > $O/r04_bench_cgfold.json 2>> $O/r04_bench.err
python3 bench.py --config config2 > $O/r04_bench_config2.json 2>> $O/r04_bench.err
python3 bench.py --config config2 --cg-fold 0 --multi-chain 0 --cpu-baseline-iters 0 > $O/r04_bench_config2_nofold.json 2>> $O/r04_bench.err
python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain-steps 6 > $O/r04_bench_config4.json 2>> $O/r04_bench.err
python3 bench.py --config config4 --dense-storage float64 --steps 10 --warmup 3 --multi-chain-steps 6 > $O/r04_bench_config4_f64.json 2>> $O/r04_bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-iters 0 > $O/r04_bench_1rank_rccl.json 2>> $O/r04_bench.err
# ---- kernel traces of the bench commands (per-launch durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_trace -- python3 bench.py --cpu-baseline-iters 0 > $O/r04_bench_under_rocprof.json 2> $O/r04_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_dense_trace -- python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain 0 --cpu-baseline-iters 0 > $O/r04_dense_bench_under_rocprof.json 2> $O/r04_dense_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r04_dense64_trace -- python3 bench.py --config config4 --dense-storage float64 --steps 10 --warmup 3 --multi-chain 0 --cpu-baseline-iters 0 > $O/r04_dense64_bench_under_rocprof.json 2> $O/r04_dense64_trace.log
# ---- HBM traffic (PMC, separate passes): the CG loop's kernels inside a chain
# (direction kernel, X~ v, X~^T w, the epilogue kernel), the K = 2
# products, the single-pass dense operator in f32 and f64 storage
for c in FETCH_SIZE WRITE_SIZE; do
  l=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --output-format csv -d $O/r04_loop_$l -- python3 scripts/iteration_traffic.py 5 60 $O/r04_iteration_run.json > $O/r04_loop_$l.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/r04_k2_$l -- python3 scripts/bench_batch_products.py config3 2 10 > $O/r04_k2_$l.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/r04_dense_$l -- python3 scripts/ab_dense_fused.py 200000 8000 5 float32 > $O/r04_dense_$l.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/r04_dense64_$l -- python3 scripts/ab_dense_fused.py 200000 8000 5 float64 > $O/r04_dense64_$l.log 2>&1
done
# ---- products on their own, register vs ring forms of the dense operator
{ python3 scripts/bench_spmv.py config3 tiled 200; python3 scripts/bench_spmv.py config2 tiled 200; } 2>&1 | grep -E "tiled geometry|avg|max abs err" > $O/r04p/r04_spmv.txt
{ for k in 2 4; do python3 scripts/bench_batch_products.py config3 $k 20; done; } 2>&1 | grep avg > $O/r04p/r04_batch_products.txt
{ for st in float32 float64; do for fl in 0 22; do BBX_DENSE_FUSED_RING=$fl python3 scripts/ab_dense_fused.py 200000 8000 20 $st; done; done; } 2>&1 | grep BBX_DENSE_FUSED_RING > $O/r04p/r04_ab_dense_fused.txt
python3 scripts/summarize_profiles.py r04 $O $O/r04p > $O/r04p/summary.log 2>&1
for sub in trace dense_trace dense64_trace loop_fetch loop_write k2_fetch k2_write dense_fetch dense_write dense64_fetch dense64_write; do rm -rf $O/r04_$sub; done
ls -la $O/r04p
