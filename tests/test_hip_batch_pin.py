"""GPU: the BATCHED draw (`bbx_batch_run`, csrc/batch.hip) pinned to the CPU
oracle column by column.

test_hip_batch.py shows a batched chain to be independent of its companions
(bit for bit) and close to its single-chain run; test_hip_chain_pin.py pins
the single-chain run to the oracle.  That leaves the batch's own machinery --
one CGState and stop flag per chain, columns that idle once their chain has
stopped, the K-column products and epilogues -- compared with the oracle only
through a chain of "approximately equal".  Here every column of a batch step is
compared with `oracle.cg_sample` (cg_sampler.py:61-94 restated) DIRECTLY: the
member chains' states are pulled before the step, each chain's normals are
regenerated from its own Philox key (`bbx_chain_eta` on the member chain: a
batch uses the chains' keys), and the oracle draws from identical inputs.  The
chains start from very different global scales, so their solves stop many CG
iterations apart: a per-chain state or stop-flag mix-up that still produced a
plausible draw would show as a wrong iteration count or a coefficient error.

Tolerances as in test_hip_chain_pin.py: coef <= 1e-6 max(1, |beta|) at equal
n_cg, the reference's 1e-5 (tests/gpu_tests/test_gibbs.py:44) otherwise;
n_cg within max(2, 10 %) of the oracle's.  The weak-shrinkage chains of these
batches need 40-60 CG iterations (the single-chain pin cases: <= 30): the
rounding difference between two summation orders grows along the recurrence,
so the equal-count bound scales as 1e-6 max(1, n_cg / 30) -- measured 1.1e-6
at 44 iterations on the K = 4 layout, whose column blocks differ most from the
oracle's row-wise sums.
"""
import math

import numpy as np
import pytest

import oracle
from oracle.gibbs import OracleGibbs
from oracle.summarizer import CoefSummarizer, regularized_prior_scale

from helpers import config2_small_problem

pytestmark = pytest.mark.gpu

ALPHA, SLAB = .5, 2.


def _sparse_case(golden_dir):
    from bayesbridge_amd import HipSparseDesignMatrix
    _, X, outcome = config2_small_problem(golden_dir)     # 20 000 x 1 000
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    return X, outcome, hip, 'logit'


def _dense_case(storage):
    from bayesbridge_amd import HipDenseDesignMatrix
    rng = np.random.default_rng(12)
    n, p = 6000, 120
    X = rng.standard_normal((n, p))
    if storage == 'float32':
        X = X.astype(np.float32).astype(np.float64)    # f32-representable
    beta = np.zeros(p)
    beta[:6] = [1.5, -1.2, 1., -.8, .7, .6]
    y = X @ beta + rng.standard_normal(n)
    hip = HipDenseDesignMatrix(X, center_predictor=True, add_intercept=True,
                               storage_dtype=storage)
    return X, y, hip, 'linear'


@pytest.mark.parametrize("kind,K", [('sparse', 2), ('sparse', 4),
                                    ('float64', 16), ('float32', 32)])
def test_every_column_of_a_batch_step_equals_the_oracle_draw(kind, K,
                                                             golden_dir):
    from bayesbridge_amd import HipChainBatch, HipGibbsChain
    if kind == 'sparse':
        X, outcome, hip, family = _sparse_case(golden_dir)
    else:
        X, outcome, hip, family = _dense_case(kind)
    n, P = hip.shape
    nu = 1
    ora = OracleGibbs(outcome, X, family, bridge_exponent=ALPHA,
                      sd_for_intercept=2. if family == 'logit' else np.inf,
                      regularizing_slab_size=SLAB)
    # chains whose solves need very different numbers of CG iterations: global
    # scales from strong to weak shrinkage, own local scales, own seeds
    gscales = np.geomspace(.004, .8, K)
    chains = []
    for c in range(K):
        if family == 'logit':
            ch = HipGibbsChain(hip, 'logit', outcome[0], n_trial=outcome[1],
                               sd_unshrunk=[2.], bridge_exponent=ALPHA,
                               slab_size=SLAB, seed=300 + 7 * c)
        else:
            ch = HipGibbsChain(hip, 'linear', outcome, sd_unshrunk=[np.inf],
                               bridge_exponent=ALPHA, slab_size=SLAB,
                               seed=300 + 7 * c)
        rng = np.random.default_rng(40 + c)
        ch.set_state(np.zeros(P), None, np.exp(rng.normal(0., 1., P - nu)),
                     float(gscales[c]))
        ch.set_gscale_update(None)       # keep the spread of the global scales
        ch.init_obs_prec()
        chains.append(ch)
    batch = HipChainBatch(chains, allow_slow=True)
    atol = 10e-6 * np.sqrt(P)
    widest_spread = 0
    for it in range(3):          # cold start, then warm starts (n_averaged 1, 2)
        draws = []
        for ch in chains:
            coef_b, obs_b, ls_b, g_b = ch.get_state()
            mean_b, square_b, n_avg = ch.get_summary()
            assert n_avg == it and ch.iteration == it
            summ = CoefSummarizer(P, nu, SLAB)
            summ.set_state({'mean': mean_b, 'square': square_b,
                            'n_averaged': n_avg})
            if family == 'linear':
                omega = obs_b * np.ones(n)
                y_gauss = outcome
            else:
                omega = obs_b
                y_gauss = (outcome[0] - outcome[1] / 2) / obs_b
            z = ora.design.Tdot(omega * y_gauss)
            prior_sd = np.concatenate((
                ora.sd_unshrunk, regularized_prior_scale(g_b, ls_b, SLAB)))
            with np.errstate(divide='ignore'):
                phi = 1 / prior_sd
            x0 = summ.extrapolate_coef_condmean(g_b, ls_b)
            sd = summ.estimate_post_sd()
            eta1, eta2 = ch.eta(it)
            draws.append(oracle.cg_sample(ora.design, omega, phi, z, x0, sd,
                                          nu, eta1, eta2, 500, atol)
                         + (summ, g_b, ls_b))
        kept, n_unconv = batch.run(1)
        assert n_unconv == 0 and batch.n_unconverged == [0] * K
        n_cg = kept['n_cg_iter'][:, 0].astype(int)
        widest_spread = max(widest_spread, int(n_cg.max() - n_cg.min()))
        for c, (coef_o, info_o, summ, g_b, ls_b) in enumerate(draws):
            coef_d = kept['coef'][c, 0]
            slack = max(2, math.ceil(.10 * info_o['n_iter']))
            assert abs(n_cg[c] - info_o['n_iter']) <= slack, \
                (it, c, n_cg[c], info_o['n_iter'])
            scale = max(1., np.abs(coef_o).max())
            tol = 1e-6 * max(1., n_cg[c] / 30.) \
                if n_cg[c] == info_o['n_iter'] else 1e-5
            if kind == 'float32' and n_cg[c] != info_o['n_iter']:
                tol = 2e-5     # f32 storage rounds the centred entries (6e-8)
            err = np.abs(coef_d - coef_o).max()
            assert err <= tol * scale, (it, c, err, n_cg[c], info_o['n_iter'])
            # the chain's own summariser saw its own draw
            summ.update(coef_d, g_b, ls_b)
            mean_a, square_a, n_avg_a = chains[c].get_summary()
            assert n_avg_a == it + 1
            assert np.abs(mean_a - summ.mean).max() <= 1e-12 * max(
                1., np.abs(summ.mean).max())
            assert np.abs(square_a - summ.square).max() <= 1e-12 * max(
                1., np.abs(summ.square).max())
            # and the log posterior of the chain's new state
            coef_a, obs_a, ls_a, g_a = chains[c].get_state()
            assert np.array_equal(coef_a, coef_d) and g_a == g_b
            lp_o = ora.logp(coef_a, g_a, obs_a)
            lp_tol = 1e-7 if kind == 'float32' else 1e-10
            assert abs(kept['logp'][c, 0] - lp_o) <= lp_tol * abs(lp_o)
    # the point of the spread of global scales: chains of one batch that stop
    # at least 5 CG iterations apart (columns idling while others iterate)
    assert widest_spread >= 5, widest_spread
    batch.close()
    for ch in chains:
        ch.close()
