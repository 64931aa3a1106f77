#!/bin/bash
# Round-6 profiles, one gpurun call (kernel traces and --pmc passes are always
# separate rocprofv3 runs).  Raw output under gpurun_out/r06_*; condensed into
# gpurun_out/r06p/ by `scripts/summarize_profiles.py r06 gpurun_out gpurun_out/r06p`
# (run at the end of this script, on the box, so that the bulky traces need not
# travel), and copied from there into profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out
mkdir -p $O/r06p
# ---- plain bench lines
python3 bench.py > $O/r06_bench.json 2> $O/r06_bench.err
python3 bench.py --steps 20 --warmup 5 > $O/r06_bench_driverflags.json 2>> $O/r06_bench.err
python3 bench.py --config config2 > $O/r06_bench_config2.json 2>> $O/r06_bench.err
python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain-steps 6 > $O/r06_bench_config4.json 2>> $O/r06_bench.err
python3 bench.py --config config4 --dense-storage float64 --steps 10 --warmup 3 --multi-chain-steps 6 > $O/r06_bench_config4_f64.json 2>> $O/r06_bench.err
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-iters 0 > $O/r06_bench_1rank_rccl.json 2>> $O/r06_bench.err
# ---- kernel traces of the bench commands (per-launch durations)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_trace -- python3 bench.py --cpu-baseline-iters 0 > $O/r06_bench_under_rocprof.json 2> $O/r06_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_dense_trace -- python3 bench.py --config config4 --steps 10 --warmup 3 --multi-chain 0 --cpu-baseline-iters 0 > $O/r06_dense_bench_under_rocprof.json 2> $O/r06_dense_trace.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_dense64_trace -- python3 bench.py --config config4 --dense-storage float64 --steps 10 --warmup 3 --multi-chain 0 --cpu-baseline-iters 0 > $O/r06_dense64_bench_under_rocprof.json 2> $O/r06_dense64_trace.log
# ---- HBM traffic (PMC, separate passes): the CG loop's kernels inside a chain
# (direction kernel, X~ v, X~^T w, the epilogue kernel), the K = 2
# products, the single-pass dense operator in f32 and f64 storage
for c in FETCH_SIZE WRITE_SIZE; do
  l=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rocprofv3 --pmc $c --output-format csv -d $O/r06_loop_$l -- python3 scripts/iteration_traffic.py 5 60 $O/r06_iteration_run.json > $O/r06_loop_$l.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/r06_dense_$l -- python3 scripts/ab_dense_fused.py 200000 8000 5 float32 > $O/r06_dense_$l.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/r06_dense64_$l -- python3 scripts/ab_dense_fused.py 200000 8000 5 float64 > $O/r06_dense64_$l.log 2>&1
done
# ---- products on their own, register vs ring forms of the dense operator
{ python3 scripts/bench_spmv.py config3 tiled 200; python3 scripts/bench_spmv.py config2 tiled 200; } 2>&1 | grep -E "tiled geometry|avg|max abs err" > $O/r06p/r06_spmv.txt
# ---- mixed designs of the reference's test shape at scale: one operator application
{ for f in 1 0; do echo "BBX_HYB_FUSED=$f"; BBX_HYB_FUSED=$f python3 scripts/bench_mixed_operator.py 1000000 2000 .9 .05 30; BBX_HYB_FUSED=$f python3 scripts/bench_mixed_operator.py 100000 10000 .9 .01 50; done; } 2>&1 | grep -E "BBX_HYB|design|operator" > $O/r06p/r06_mixed_operator.txt
python3 scripts/summarize_profiles.py r06 $O $O/r06p > $O/r06p/summary.log 2>&1
for sub in trace dense_trace dense64_trace loop_fetch loop_write dense_fetch dense_write dense64_fetch dense64_write; do rm -rf $O/r06_$sub; done
ls -la $O/r06p
