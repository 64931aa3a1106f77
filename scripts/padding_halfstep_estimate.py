"""What would a half-width last step buy?  (LABNOTES R6.6 (c); CPU only.)
For the X orientation of a config-3-like design (1M x 50k binary, f = .002):
row segments per (panel of PR rows, column block of W columns), sorted by
length inside a tile, 128 rows per slice (two per lane), groups of five per
step.  Counts the 16-byte lane-steps stored today and with an 8-byte last
step for slices whose rows all have at most `half` entries left in it.
    python scripts/padding_halfstep_estimate.py [n] [p] [f]"""
import sys

import numpy as np

sys.path.insert(0, 'bayes-bridge_amd')
from bayesbridge_amd import simulate

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
f = float(sys.argv[3]) if len(sys.argv) > 3 else .002
W, PR = 12544, 3968            # the geometry of DESIGN.md 2 for this shape
X = simulate.simulate_binary_csr_fast(n, p, f, seed=111)
print("nnz", X.nnz)
block = (X.indices // W).astype(np.int8)
n_block = int(block.max()) + 1
rows = np.repeat(np.arange(n), np.diff(X.indptr))
# entries per (row, block)
seg = np.zeros((n, n_block), dtype=np.int32)
np.add.at(seg, (rows, block), 1)
tot_steps = tot_half = entries = 0
for r0 in range(0, n, PR):
    for b in range(n_block):
        ln = np.sort(seg[r0:r0 + PR, b])[::-1]
        ln = ln[ln > 0]
        entries += int(ln.sum())
        for s0 in range(0, len(ln), 128):
            sl = ln[s0:s0 + 128]
            steps = -(-int(sl[0]) // 5)
            tot_steps += steps
            # entries left for the last step, per row
            left = np.maximum(sl - 5 * (steps - 1), 0)
            tot_half += steps - (0.5 if left.max() <= 2 else 0.)
lane_bytes = 16 * 64
useful = entries * 1.6
for name, st in (("today", tot_steps), ("half last step", tot_half)):
    b = st * lane_bytes
    print("%-16s %.1f MB of ids, padding %.2f %%" % (name, b / 1e6,
                                                      100 * (1 - useful / b)))
