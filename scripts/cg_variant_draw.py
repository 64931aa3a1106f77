"""One CG draw through bbx_cg_sample on a seeded problem, saved to an .npz, for
comparing build/run-time variants of the loop that are selected once per
process (BBX_CG_MERGE_UPDATE=0|1, BBX_CG_NO_SKIP, BBX_CG_FUSED):
    python scripts/cg_variant_draw.py out.npz [sparse|dense] [n] [p] [seed] [maxiter]
sparse: tiled layout, mixed binary/valued columns; dense: f32 storage (the
single-pass operator kernel)."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bayesbridge_amd import HipCGSampler, HipDenseDesignMatrix, \
    HipSparseDesignMatrix
from helpers import cg_inputs, mixed_design

out = sys.argv[1]
kind = sys.argv[2] if len(sys.argv) > 2 else "sparse"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
p = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
maxiter = int(sys.argv[6]) if len(sys.argv) > 6 else 500
if kind == "sparse":
    X = mixed_design(n, p, binary_frac=.8, seed=seed)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
else:
    X = np.random.default_rng(seed).standard_normal((n, p)).astype(np.float32)
    hip = HipDenseDesignMatrix(X.astype(np.float64), center_predictor=True,
                               add_intercept=True, storage_dtype='float32')
n, P = hip.shape
# (a narrow spread of prior scales: a few dozen iterations, like the chain's
# solves; the wide default takes hundreds, where any two roundings of the same
# recurrence drift apart by 1e-7)
inp = cg_inputs(n, P, seed=seed, lam_log_sd=.3)
hip.reset_matvec_count()
warnings.simplefilter('ignore')       # short runs stop at maxiter on purpose
coef, info = HipCGSampler(inp['n_unshrunk']).sample(
    hip, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
    coef_cg_init=inp['coef_cg_init'], precond_by='prior',
    coef_scaled_sd=inp['coef_scaled_sd'], maxiter=maxiter,
    atol=10e-6 * np.sqrt(P), seed=seed + 7)
np.savez(out, coef=coef, n_iter=info['n_iter'],
         counts=np.array(hip.get_dot_count()))
print("n_iter %d converged %s counts %s" % (info['n_iter'], info['converged'],
                                            hip.get_dot_count()))
