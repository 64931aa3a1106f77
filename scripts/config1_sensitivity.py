"""How far do 20 iterations of the exact-seed config-1 chain (dense 2000 x 500,
linear, cg; BASELINE configs[0]) move under perturbations at rounding level?
CPU only: the oracle chain (bit-identical to the reference on this config,
tests/test_oracle_vs_reference.py) is run on X and on eight copies of X whose
entries are perturbed by 1e-15 (relative, random) -- the size of a re-ordered
sum -- with the same seed.  Prints, per perturbed run, max |coef_last - base|
and the largest shift of a CG stopping iteration.  Basis of the bounds in
tests/test_hip_chain.py::test_dense_linear_chain_config1_summary.
    python scripts/config1_sensitivity.py > profiles/r06_config1_sensitivity.txt"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
from oracle.gibbs import OracleGibbs
from bayesbridge_amd import simulate

warnings.simplefilter('ignore')
np.random.seed(111)
X = np.random.randn(2000, 500)
beta = simulate.demo_beta(500)
y = simulate.simulate_outcome(X, beta, 'linear', seed=1)
kw = dict(bridge_exponent=.5, regularizing_slab_size=2.)


def run(Xp):
    return OracleGibbs(y, Xp, 'linear', **kw).gibbs(
        20, seed=111, init={'global_scale': .01})


base = run(X)
print("# scripts/config1_sensitivity.py: exact-seed config-1 chain (oracle, 20 "
      "iterations), X perturbed by 1e-15 relative")
print("base n_cg:", base['n_cg_iter'].astype(int).tolist())
prng = np.random.default_rng(7)
worst_c, worst_n, worst_g = 0., 0, 0.
for k in range(8):
    Xp = X * (1 + 1e-15 * prng.standard_normal(X.shape))
    out = run(Xp)
    dc = np.abs(out['coef'] - base['coef']).max(axis=0)
    dn = np.abs(out['n_cg_iter'] - base['n_cg_iter'])
    dg = np.abs(out['global_scale'] / base['global_scale'] - 1)
    first = int(np.argmax(dc > 1e-9)) if (dc > 1e-9).any() else -1
    print("perturbation %d: max|coef - base| per iteration: first > 1e-9 at "
          "iteration %d; last sample %.2e; max over the run %.2e; n_cg shift "
          "max %d; tau rel. shift max %.1e"
          % (k, first, dc[-1], dc.max(), int(dn.max()), dg.max()))
    worst_c, worst_n, worst_g = max(worst_c, dc.max()), \
        max(worst_n, int(dn.max())), max(worst_g, dg.max())
print("worst over 8 perturbations: coef %.2e, n_cg shift %d, tau rel %.1e"
      % (worst_c, worst_n, worst_g))
