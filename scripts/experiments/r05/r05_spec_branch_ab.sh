#!/bin/bash
# A/B: tau / lambda branch enqueued by the CG loop behind its stop test
# (BBX_CHAIN_SPEC_BRANCH, default 1) x wave priority for the lambda kernel
# (BBX_LSCALE_PRIO, default 1)
out=gpurun_out/${1:-r05g}; mkdir -p $out
cfg=${2:-config3}
for rep in 1 2 3; do for v in 11 01 10 00; do
  export BBX_CHAIN_SPEC_BRANCH=${v:0:1} BBX_LSCALE_PRIO=${v:1:1}
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/sb_${cfg}_${v}_$rep.json 2> $out/sb_${cfg}_${v}_$rep.err
  python3 -c "import json;d=json.load(open('$out/sb_${cfg}_${v}_$rep.json'));print('$cfg spec,prio=$v rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
