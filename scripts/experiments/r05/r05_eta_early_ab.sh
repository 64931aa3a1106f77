#!/bin/bash
# A/B: next draw's normals filled on the second stream at the START of the solve
# (chain_head_hook; BBX_ETA_AHEAD unset) / in front of the Polya-Gamma kernel (=1,
# the placement of R5.8) / not ahead at all (=0); x lambda-kernel priority
out=gpurun_out/${1:-r05i}; mkdir -p $out
cfg=${2:-config3}
for rep in 1 2 3; do for v in e1 e0 11 10 01; do
  if [ ${v:0:1} = e ]; then unset BBX_ETA_AHEAD; else export BBX_ETA_AHEAD=${v:0:1}; fi
  export BBX_LSCALE_PRIO=${v:1:1}
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/ee_${cfg}_${v}_$rep.json 2> $out/ee_${cfg}_${v}_$rep.err
  python3 -c "import json;d=json.load(open('$out/ee_${cfg}_${v}_$rep.json'));print('$cfg eta,lprio=$v rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
