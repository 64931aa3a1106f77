#!/bin/bash
# kernel-trace averages of the sampler kernels: working tree against ./ab_ref
# (built by r05_ab_against_ref.sh, run that first)
root=$PWD
out=$root/gpurun_out/${1:-r05w}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $root
for v in work ref; do
  if [ $v = ref ]; then export BBX_PACKAGE_DIR=$root/ab_ref/bayes-bridge_amd; else unset BBX_PACKAGE_DIR; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/tr_$v -- python3 bench.py --steps 30 --warmup 5 --repeat 1 --cpu-baseline-iters 0 --multi-chain 0 > $out/tr_$v.json 2> $out/tr_$v.err
  f=$(find $out/tr_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v (name, calls, avg ns, min, max)"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    n = row.get('Name', '')
    if any(k in n for k in ('lscale', 'chain_pg', 'gscale')):
        print(n.split('(')[0][-40:], row.get('Calls'), row.get('AverageNs'), row.get('MinNs'), row.get('MaxNs'))
PY
  rm -rf $out/tr_$v
done
