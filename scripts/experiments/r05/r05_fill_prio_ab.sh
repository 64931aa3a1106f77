# A/B: wave priority for the two fill kernels of the next draw's normals (BBX_FILL_PRIO=1 default / 0)
mkdir -p gpurun_out/r05r
for rep in 1 2 3; do for v in 1 0; do
  BBX_FILL_PRIO=$v python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05r/fp${v}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05r/fp${v}_$rep.json'));print('fill_prio=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
