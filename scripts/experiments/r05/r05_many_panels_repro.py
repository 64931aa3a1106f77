"""Repro: a device chain on a tiled design whose X layout has more row panels
than NPART (256) -- any design past ~1M rows; forced here with BBX_TILED_PR."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np

from bayesbridge_amd import HipGibbsChain, HipSparseDesignMatrix, simulate

intercept = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
n, p = 100000, 3000
X = simulate.simulate_binary_csr_fast(n, p, .02, seed=3)
hip = HipSparseDesignMatrix(X, add_intercept=intercept, storage='tiled')
print("intercept", intercept, "tiled", hip.tiled_info())
y = (np.random.default_rng(0).random(n) < .3).astype(np.float64)
chain = HipGibbsChain(hip, 'logit', y, sd_unshrunk=[2.] if intercept else [],
                      slab_size=1., seed=3)
if len(sys.argv) > 2:      # a sensible start, as bench.py's
    P = hip.shape[1]
    chain.set_state(np.zeros(P), None, np.ones(P - int(intercept)), .01)
    chain.init_obs_prec()
out, bad = chain.run(4)
print("n_cg", out['n_cg_iter'], "unconverged", bad, "finite",
      bool(np.all(np.isfinite(out['coef']))))
