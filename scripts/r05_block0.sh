#!/bin/bash
# Round 5: where does the block-0 bias of `value` come from?  
# A/B of bench.py with the kernel stamps on in the first block only (what the
# line reports), in all blocks, in none.  Output: gpurun_out/$1/*.json
out=gpurun_out/${1:-r05a}
mkdir -p $out
for cfg in config3 config2 config4; do
  for tb in first all none; do
    for rep in 1 2; do
      python bench.py --config $cfg --timing-blocks $tb --cpu-baseline-iters 0 \
        --multi-chain 0 > $out/b0_${cfg}_${tb}_$rep.json 2> $out/b0_${cfg}_${tb}_$rep.err
    done
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$out/b0_*.json")):
    try:
        d = json.load(open(f))
        r = d["repeat"]
        print(f.split("/")[-1], d["value"], r["values"], r["median"],
              "b0/median=%.4f" % (r["values"][0] / r["median"]))
    except Exception as e:
        print(f, "ERR", e)
PY
