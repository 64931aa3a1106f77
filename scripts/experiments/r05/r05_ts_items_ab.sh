# lambda kernel: coefficients per 256-thread block and pass (BBX_TS_ITEMS; default 128 at p = 50k)
mkdir -p gpurun_out/r05o
for rep in 1 2; do for v in 128 32 64 256 16; do
  BBX_TS_ITEMS=$v python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05o/items${v}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05o/items${v}_$rep.json'));print('items=$v', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
