# A/B: Polya-Gamma kernel launched before the tau / lambda branch when the pass is already under way
# (BBX_PG_FIRST=1 default / 0), with the next draw's normals in front of it (BBX_ETA_WHERE=0) or behind
# the lambda kernel (1)
mkdir -p gpurun_out/r05t
for rep in 1 2 3; do for v in "1 0" "1 1" "0 0"; do set -- $v
  BBX_PG_FIRST=$1 BBX_ETA_WHERE=$2 python3 bench.py --config config3 --steps 50 --warmup 10 --repeat 3 --cpu-baseline-iters 0 --multi-chain 0 > gpurun_out/r05t/pf$1_w$2_$rep.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r05t/pf$1_w$2_$rep.json'));print('pg_first=$1 eta_where=$2', d['value'], d['repeat']['values'], d['repeat']['us_per_cg_iter'])"
done; done
