#!/bin/bash
# Round 5: Polya-Gamma draws in rounds (chain.hip polya_gamma_block): duration of
# chain_pg_kernel in the config-3 chain for E = 1, 2, 4, 8 draws per lane and pass.
#   bash scripts/r05_pg_ab.sh <outdir under gpurun_out>
out=gpurun_out/${1:-r05d}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for e in 1 2 4 8; do
  rm -rf $out/prof_e$e
  BBX_PG_ITEMS=$e timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_e$e -o run -- \
    python3 bench.py --config config3 --steps 30 --warmup 5 --burnin 60 --repeat 1 \
      --cpu-baseline-iters 0 --multi-chain 0 > $out/bench_e$e.json 2> $out/bench_e$e.err
  f=$(find $out/prof_e$e -name "*kernel_stats.csv" | head -1)
  echo "== E=$e: $(python3 -c "import json;d=json.load(open('$out/bench_e$e.json'));print(d['value'], d['config']['mean_n_cg_iter'])")"
  if [ -n "$f" ]; then
    grep -E "chain_pg_kernel|chain_lscale_kernel|Name" "$f" | cut -c1-200
    cp "$f" $out/kernel_stats_e$e.csv
  else
    echo "no kernel_stats.csv under $out/prof_e$e"; ls -R $out/prof_e$e | head
  fi
  rm -rf $out/prof_e$e
done
