# A/B of the stop-flag look of the CG loop: blocking hipStreamSynchronize (0) against
# polling hipStreamQuery (1).  Usage: bash r05_spin_ab.sh (on the GPU box)
mkdir -p gpurun_out/r05h
for rep in 1 2 3; do for cfg in config2 config3; do for sp in 0 1; do
  BBX_CG_SPIN=$sp python3 bench.py --config $cfg --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05h/spin${sp}_${cfg}_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05h/spin${sp}_${cfg}_$rep.json'));print('$cfg spin=$sp', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done; done
