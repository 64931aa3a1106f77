# A/B: where the next draw's normals are filled (BBX_ETA_WHERE=0 in front of the PG kernel on the
# design's stream / 1 behind the lambda kernel on the branch's stream), both with wave priority
mkdir -p gpurun_out/r05s
for rep in 1 2 3; do for v in 0 1; do
  BBX_ETA_WHERE=$v python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05s/w${v}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05s/w${v}_$rep.json'));print('eta_where=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
