#!/bin/bash
# A/B: Polya-Gamma kernel with E elements per lane (csrc/pg_queue.hpp; BBX_PG_ELEMS
# unset = by size, 8 at config 3) against the one-lane kernel of rounds 1-4 (0)
out=gpurun_out/${1:-r05f}; mkdir -p $out
cfg=${2:-config3}
for rep in 1 2 3; do for e in auto 0 4; do
  if [ $e = auto ]; then unset BBX_PG_ELEMS; else export BBX_PG_ELEMS=$e; fi
  python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 \
     --cpu-baseline-iters 0 --multi-chain 0 > $out/pgq_${cfg}_${e}_$rep.json 2> $out/pgq_${cfg}_${e}_$rep.err
  python3 -c "import json;d=json.load(open('$out/pgq_${cfg}_${e}_$rep.json'));print('$cfg E=$e rep $rep', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
