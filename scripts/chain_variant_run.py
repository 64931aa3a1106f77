"""A short device chain (bbx_chain_run) on a seeded problem, samples saved to
an .npz -- for comparing variants that are selected once per process
(BBX_CHAIN_FORK=0|1, BBX_CG_MERGE_RESID=0|1, BBX_CG_MERGE_UPDATE=0|1):
    python scripts/chain_variant_run.py out.npz [logit|linear] [n] [p] [iters] [binary_frac] [freq]
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
from bayesbridge_amd import HipSparseDesignMatrix, simulate
from bayesbridge_amd.device_chain import HipGibbsChain

out = sys.argv[1]
family = sys.argv[2] if len(sys.argv) > 2 else "logit"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
p = int(sys.argv[4]) if len(sys.argv) > 4 else 300
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 6
binary_frac = float(sys.argv[6]) if len(sys.argv) > 6 else .8
freq = float(sys.argv[7]) if len(sys.argv) > 7 else .1
warnings.simplefilter("ignore")
big = binary_frac == 1. and n >= 200000
if big:
    # (generated in HBM like bench.py's designs: the exact replay of the
    # reference's generator draws one column at a time; the outcome is then
    # independent of X -- these runs compare variants of ONE chain bit for bit)
    import torch
    indptr, indices = simulate.simulate_binary_csr_device(n, p, freq, seed=3)
    nnz = int(indices.numel())
    offset = torch.bincount(indices.long(), minlength=p).double() / n
    torch.cuda.synchronize()
    hip = HipSparseDesignMatrix.from_device_csr(
        n, p, nnz, indptr.data_ptr(), indices.data_ptr(), None,
        offset.data_ptr(), add_intercept=True, storage='tiled')
    del indptr, indices
    rng_y = np.random.default_rng(4)
    y = ((rng_y.random(n) < .3).astype(np.float64), np.ones(n)) \
        if family == "logit" else rng_y.standard_normal(n)
else:
    X = simulate.simulate_design_csr(n, p, binary_frac=binary_frac,
                                     binary_pred_freq=freq, seed=3)
    beta = np.zeros(p)
    beta[:5], beta[5:10] = 1.5, -1.
    y = simulate.simulate_outcome(X, beta, family, seed=4)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
P = hip.shape[1]
if family == "logit":
    chain = HipGibbsChain(hip, 'logit', y[0], n_trial=y[1], sd_unshrunk=[2.],
                          bridge_exponent=.5, slab_size=2., gscale_shape=1.5,
                          gscale_rate=.3, seed=17)
else:
    chain = HipGibbsChain(hip, 'linear', y, sd_unshrunk=[np.inf],
                          bridge_exponent=.5, slab_size=2., seed=17)
rng = np.random.default_rng(5)
chain.set_state(np.zeros(P), None, np.exp(rng.normal(0., 1., P - 1)), .07)
chain.init_obs_prec()
kept, n_unconv = chain.run(iters, save=('coef', 'local_scale', 'obs_prec'))
np.savez(out, coef=kept['coef'], local_scale=kept['local_scale'],
         obs_prec=kept['obs_prec'], global_scale=kept['global_scale'],
         logp=kept['logp'], n_cg_iter=kept['n_cg_iter'])
print("n_cg", kept['n_cg_iter'].tolist(), "unconverged", n_unconv,
      "format", hip.storage_format, "cg launches", hip.cg_launches,
      "naps", hip.cg_stats()[2])
