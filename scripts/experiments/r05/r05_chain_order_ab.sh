# A/B of the launch order after a draw (BBX_CHAIN_ORDER=0 committed: pass, branch, PG;
# 1: branch first, then pass, then PG), config 3, alternating, us per CG iteration and it/s
mkdir -p gpurun_out/r05j
for rep in 1 2 3; do for o in 0 1; do
  BBX_CHAIN_ORDER=$o python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05j/order${o}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05j/order${o}_$rep.json'));print('order=$o', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
