# A/B: the next draw's normals filled under the tau / lambda branch (BBX_ETA_AHEAD=1 default / 0)
mkdir -p gpurun_out/r05m
for rep in 1 2 3; do for cfg in config3 config2; do for v in 1 0; do
  BBX_ETA_AHEAD=$v python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05m/eta${v}_${cfg}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05m/eta${v}_${cfg}_$rep.json'));print('$cfg eta_ahead=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
