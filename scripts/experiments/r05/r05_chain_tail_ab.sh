# A/B: the pass for X~ beta enqueued by the CG loop behind the stop test (BBX_CHAIN_TAIL=1 default / 0)
mkdir -p gpurun_out/r05n
for rep in 1 2 3; do for cfg in config3 config2; do for v in 1 0; do
  BBX_CHAIN_TAIL=$v python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05n/tail${v}_${cfg}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05n/tail${v}_${cfg}_$rep.json'));print('$cfg tail=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
