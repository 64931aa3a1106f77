"""Generates tests/golden/chain_logit_binary_100000x10000_first10.npz by
importing the upstream reference (build container only; see ref_import.py):
BASELINE config 2 at FULL size -- simulate_design(100000, 10000, binary_frac=1,
binary_pred_freq=.01, format_='sparse', seed=111), demo coefficients and prior,
gibbs(10, init={'global_scale': .01}, coef_sampler_type='cg', seed=111).
The fixture is DATA: checksums of the design and outcome (the tests regenerate
both with bayesbridge_amd.simulate and compare), the reference's first 10
samples of the scalars and of 256 coefficients (the first 64, which hold the 15
signals, and 192 drawn at random), and its n_cg_iter.  ~60 s for the literal
simulate_design (8 GB transient), ~15 s for the chain.

    python tests/golden/make_config2_full.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_import  # noqa: E402

warnings.simplefilter('ignore')
bb, refsim = ref_import.import_reference()
from bayesbridge import BayesBridge, RegressionModel, RegressionCoefPrior  # noqa

n, p, f = 100000, 10000, .01
X = refsim.simulate_design(n, p, binary_frac=1., binary_pred_freq=f,
                           format_='sparse', seed=111)
X = X.tocsr()
X.sort_indices()
beta = np.zeros(p)
beta[:5], beta[5:10], beta[10:15] = 1.5, 1., .5
y = refsim.simulate_outcome(X, beta, 'logit', seed=1)
bridge = BayesBridge(
    RegressionModel(y, X, 'logit'),
    RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.))
samples, info = bridge.gibbs(10, 0, init={'global_scale': .01},
                             coef_sampler_type='cg', seed=111)
coef = samples['coef']
rng = np.random.default_rng(5)
picked = np.concatenate([np.arange(64),
                         np.sort(rng.choice(np.arange(64, p + 1), 192,
                                            replace=False))])
n_success, n_trial = y
np.savez_compressed(
    os.path.join(HERE, 'chain_logit_binary_100000x10000_first10.npz'),
    shape=np.array([n, p]), freq=f, nnz=X.nnz,
    indices_checksum=np.int64(
        (X.indices.astype(np.int64) * (np.arange(X.nnz) % 1009 + 1)).sum()),
    indptr_tail=X.indptr[-4:], n_success_sum=n_success.sum(),
    n_success_head=n_success[:32], n_trial_head=n_trial[:32],
    picked=picked, coef_first10=coef[picked, :10],
    coef_abs_sum_first10=np.abs(coef[:, :10]).sum(axis=0),
    global_scale_first10=samples['global_scale'][:10],
    logp_first10=samples['logp'][:10],
    n_cg_iter=info['_reg_coef_sampling_info']['n_cg_iter'])
print('config 2 (full size) written: nnz', X.nnz, 'n_cg',
      info['_reg_coef_sampling_info']['n_cg_iter'])
