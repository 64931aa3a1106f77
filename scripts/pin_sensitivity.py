"""How far does the stopping iteration of a CG draw move under perturbations
at rounding level?  Replays tests/test_hip_chain_pin.py's loop for one case
and, at every iteration, runs the CPU oracle on the device state with Omega
perturbed by 1e-15 (relative, random) eight times; prints the oracle's range
of n_iter next to the device's count.
    python scripts/pin_sensitivity.py [family] [kind] [storage]"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle
from oracle.gibbs import OracleGibbs
from oracle.summarizer import CoefSummarizer, regularized_prior_scale
import test_hip_chain_pin as T
from bayesbridge_amd.device_chain import HipGibbsChain

family = sys.argv[1] if len(sys.argv) > 1 else 'linear'
kind = sys.argv[2] if len(sys.argv) > 2 else 'sparse'
storage = sys.argv[3] if len(sys.argv) > 3 else 'tiled'
warnings.simplefilter('ignore')
X, y = T._problem(family, kind)
hip = T._designs(X, kind, storage)
if family == 'logit':
    outcome = y
    chain = HipGibbsChain(hip, 'logit', y[0], n_trial=y[1], sd_unshrunk=[2.],
                          bridge_exponent=T.ALPHA, slab_size=T.SLAB,
                          gscale_shape=1.5, gscale_rate=.3, seed=17)
else:
    outcome = y
    chain = HipGibbsChain(hip, 'linear', y, sd_unshrunk=[np.inf],
                          bridge_exponent=T.ALPHA, slab_size=T.SLAB, seed=17)
ora = OracleGibbs(outcome, X, family, bridge_exponent=T.ALPHA,
                  sd_for_intercept=2. if family == 'logit' else np.inf,
                  regularizing_slab_size=T.SLAB,
                  gscale_shape=1.5 if family == 'logit' else 0.,
                  gscale_rate=.3 if family == 'logit' else 0.)
n, P = hip.shape
rng = np.random.default_rng(5)
chain.set_state(np.zeros(P), None, np.exp(rng.normal(0., 1., P - 1)), .07)
chain.init_obs_prec()
atol = 10e-6 * np.sqrt(P)
prng = np.random.default_rng(99)
for it in range(4):
    coef_b, obs_b, ls_b, g_b = chain.get_state()
    mean_b, square_b, n_avg = chain.get_summary()
    summ = CoefSummarizer(P, 1, T.SLAB)
    summ.set_state({'mean': mean_b, 'square': square_b, 'n_averaged': n_avg})
    if family == 'linear':
        omega, y_gauss = obs_b * np.ones(n), y
    else:
        omega, y_gauss = obs_b, (outcome[0] - outcome[1] / 2) / obs_b
    z = ora.design.Tdot(omega * y_gauss)
    prior_sd = np.concatenate((ora.sd_unshrunk,
                               regularized_prior_scale(g_b, ls_b, T.SLAB)))
    with np.errstate(divide='ignore'):
        phi = 1 / prior_sd
    x0 = summ.extrapolate_coef_condmean(g_b, ls_b)
    sd = summ.estimate_post_sd()
    eta1, eta2 = chain.eta(it)
    counts = []
    for k in range(9):
        om = omega if k == 0 else omega * (1 + 1e-15 * prng.standard_normal(n))
        _, info = oracle.cg_sample(ora.design, om, phi, z, x0, sd, 1, eta1,
                                   eta2, 500, atol)
        counts.append(info['n_iter'])
    kept, _ = chain.run(1, save=('coef',))
    print("iteration %d: device n_cg %d, oracle %d, oracle under 1e-15 "
          "perturbations of Omega %d..%d" % (
              it, int(kept['n_cg_iter'][0]), counts[0], min(counts),
              max(counts)))
