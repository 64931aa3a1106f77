#!/bin/bash
# A/B of the single-pass dense operator: register prefetch vs LDS-DMA ring
# (scripts/ab_dense_fused.py), one process per variant and shape.
for shape in "200000 8000 20" "20001 4001 5" "4099 801 5" "30000 5000 5"; do
  for ring in 0 22; do
    BBX_DENSE_FUSED_RING=$ring timeout 300 python scripts/ab_dense_fused.py $shape 2>&1 | tail -1
  done
done
