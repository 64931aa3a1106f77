# A/B: s_setprio 3 in the three small kernels that lead the tau / lambda branch
# (BBX_BRANCH_PRIO=1 default / 0), configs 3 and 2, alternating
mkdir -p gpurun_out/r05l
for rep in 1 2 3; do for cfg in config3 config2; do for v in 1 0; do
  BBX_BRANCH_PRIO=$v python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05l/prio${v}_${cfg}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05l/prio${v}_${cfg}_$rep.json'));print('$cfg prio=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
