#!/bin/bash
# chain it/s at config 3 with the one-lane-per-draw PG kernel (0) and the rounds (E)
out=gpurun_out/${1:-r05d}; mkdir -p $out
for rep in 1 2; do for e in 0 4 8 2; do
  BBX_PG_ITEMS=$e python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/ab2_e${e}_$rep.json 2> $out/ab2_e${e}_$rep.err
  python3 -c "import json;d=json.load(open('$out/ab2_e${e}_$rep.json'));print('E=$e rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
