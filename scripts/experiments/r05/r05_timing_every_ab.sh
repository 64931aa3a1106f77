#!/bin/bash
# What the kernel stamps of the timed region cost: one launch in 16 / 64 / 256 / none
out=gpurun_out/${1:-r05m}; mkdir -p $out
for cfg in config3 config2; do for rep in 1 2 3; do for e in 16 64 256 none; do
  if [ $e = none ]; then fl="--timing-blocks none"; else fl="--timing-every $e"; fi
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 $fl \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/te_${cfg}_${e}_$rep.json 2> $out/te_${cfg}_${e}_$rep.err
  python3 -c "import json;d=json.load(open('$out/te_${cfg}_${e}_$rep.json'));r=d.get('roofline') or {};o=r.get('other') or {};print('$cfg every=$e rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'], {k:(v['launches'],v['avg_ms']) for k,v in o.items()})"
done; done; done
