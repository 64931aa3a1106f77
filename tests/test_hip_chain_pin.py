"""GPU: exact pin of the device-resident chain that bench.py times
(`bbx_chain_run`, Philox streams) against the CPU oracle.

Philox is counter-based, so the normals a chain iteration consumes can be
regenerated (`bbx_chain_eta`).  Feeding them to the oracle makes every
deterministic piece of an iteration comparable on identical inputs:

  chain_prior_kernel     phi, x0, sd, z        reg_coef_sampler.py:74-89
  CG draw                coef, n_iter          cg_sampler.py:20-94
  chain_summary_kernel   mean, square          reg_coef_posterior_summarizer.py:93-124
  chain_gscale_kernel    logp (+ 'optimize' / fixed tau)   bayesbridge.py:412-456,480-511
  fill_normal_kernel     eta1, eta2            cg_sampler.py:61-62 (distribution)

The random draws themselves (Polya-Gamma, tilted stable, Gamma) are
distribution-tested in test_hip_chain.py.
"""
import math
import warnings

import numpy as np
import pytest
import scipy.sparse as sparse
from scipy import stats

import oracle
from oracle.gibbs import OracleGibbs, unit_bridge_magnitude
from oracle.summarizer import CoefSummarizer, regularized_prior_scale

pytestmark = pytest.mark.gpu

ALPHA, SLAB = .5, 2.


def _problem(family, kind, n=3000, p=200, seed=3, binary=False):
    from bayesbridge_amd import simulate
    if kind == 'dense':
        rng = np.random.default_rng(seed)
        X = rng.standard_normal((n, p))
        if family == 'linear':
            X = X.astype(np.float32).astype(np.float64)   # f32-representable
    else:
        X = simulate.simulate_design_csr(n, p,
                                         binary_frac=1. if binary else .8,
                                         binary_pred_freq=.1, seed=seed)
    beta = np.zeros(p)
    beta[:5], beta[5:10] = 1.5, -1.
    y = simulate.simulate_outcome(X, beta, family, seed=seed + 1)
    return X, y


def _designs(X, kind, storage):
    from bayesbridge_amd import HipDenseDesignMatrix, HipSparseDesignMatrix
    if kind == 'dense':
        hip = HipDenseDesignMatrix(X, center_predictor=True,
                                   add_intercept=True, storage_dtype=storage)
    else:
        hip = HipSparseDesignMatrix(X, center_predictor=True,
                                    add_intercept=True,
                                    storage=storage.split('+')[0])
        if storage.endswith('+fold'):
            # the opt-in 3-launch CG iteration (direction step inside the
            # X~ v kernel, csrc/common.hpp DotFold): same recurrence
            hip.set_cg_fold(True)
            assert hip.cg_launches == 3
    return hip


CASES = [('logit', 'sparse', 'tiled'), ('logit', 'sparse', 'csr'),
         ('logit', 'sparse', 'tiled+fold'), ('linear', 'sparse', 'tiled+fold'),
         ('linear', 'dense', 'float64'), ('linear', 'dense', 'float32'),
         ('linear', 'sparse', 'tiled'), ('logit', 'dense', 'float64')]


@pytest.mark.parametrize("family,kind,storage", CASES)
def test_device_chain_iteration_equals_oracle(family, kind, storage):
    """Four consecutive iterations of bbx_chain_run, each compared with the
    oracle started from the device state before it (so n_averaged = 0..3
    exercises both branches of the sd estimate and the cold/warm CG start)."""
    from bayesbridge_amd.device_chain import HipGibbsChain
    # (the folded direction step applies to value-free layouts: binary design)
    X, y = _problem(family, kind, binary=storage.endswith('+fold'))
    hip = _designs(X, kind, storage)
    if kind == 'sparse' and storage == 'tiled':
        # binary_frac = .8 of 200 columns: the 39 continuous ones are a dense
        # block that every operator application of the CG loop takes in ONE pass
        # (csrc/spmv_tiled.hip hyb_dense_fused_kernel; up to 8 columns ride in
        # the value-free kernel's epilogue instead)
        assert hip.hybrid_info is not None
        assert hip.hybrid_info['dense_cols'] == 39
        assert hip.hybrid_info['rest_nnz'] == 0
    if family == 'logit':
        n_success, n_trial = y
        outcome = (n_success, n_trial)
        chain = HipGibbsChain(hip, 'logit', n_success, n_trial=n_trial,
                              sd_unshrunk=[2.], bridge_exponent=ALPHA,
                              slab_size=SLAB, gscale_shape=1.5,
                              gscale_rate=.3, seed=17)
    else:
        outcome = y
        chain = HipGibbsChain(hip, 'linear', y, sd_unshrunk=[np.inf],
                              bridge_exponent=ALPHA, slab_size=SLAB, seed=17)
    ora = OracleGibbs(outcome, X, family, bridge_exponent=ALPHA,
                      sd_for_intercept=2. if family == 'logit' else np.inf,
                      regularizing_slab_size=SLAB,
                      gscale_shape=1.5 if family == 'logit' else 0.,
                      gscale_rate=.3 if family == 'logit' else 0.)
    n, P = hip.shape
    nu = 1
    rng = np.random.default_rng(5)
    coef0 = np.zeros(P)
    chain.set_state(coef0, None, np.exp(rng.normal(0., 1., P - nu)), .07)
    chain.init_obs_prec()
    atol = 10e-6 * np.sqrt(P)
    hip.reset_matvec_count()
    for it in range(4):
        coef_b, obs_b, ls_b, g_b = chain.get_state()
        mean_b, square_b, n_avg = chain.get_summary()
        assert n_avg == it and chain.iteration == it
        # ---- oracle side of reg_coef_sampler.py:74-89 from the same state
        summ = CoefSummarizer(P, nu, SLAB)
        summ.set_state({'mean': mean_b, 'square': square_b,
                        'n_averaged': n_avg})
        if family == 'linear':
            omega = obs_b * np.ones(n)
            y_gauss = y
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
        eta1, eta2 = chain.eta(it)
        coef_o, info_o = oracle.cg_sample(ora.design, omega, phi, z, x0, sd,
                                          nu, eta1, eta2, 500, atol)
        # ---- one device iteration
        before = hip.get_dot_count()
        kept, n_unconv = chain.run(1, save=('coef', 'local_scale', 'obs_prec'))
        assert n_unconv == 0
        coef_d = kept['coef'][0]
        n_cg = int(kept['n_cg_iter'][0])
        # +-2 around the oracle's stopping iteration; solves of 40-100
        # iterations sit on a flat stretch of the residual curve where rounding
        # moves the stop by several iterations: the ORACLE's own count ranges
        # over 75..82 at iteration 3 of the linear/sparse case when Omega is
        # perturbed by 1e-15 (scripts/pin_sensitivity.py,
        # profiles/r02_pin_sensitivity.txt), the device's summation order
        # gives 74.  Hence +-10 %; the coefficients are compared below.
        slack = max(2, math.ceil(.10 * info_o['n_iter']))
        assert abs(n_cg - info_o['n_iter']) <= slack, (n_cg, info_o['n_iter'])
        scale = max(1., np.abs(coef_o).max())
        tol = 1e-6 if n_cg == info_o['n_iter'] else 1e-5
        assert np.abs(coef_d - coef_o).max() <= tol * scale, \
            (it, np.abs(coef_d - coef_o).max())
        # counters: n_cg operator applications; X~ (s x0) for a warm start; ONE
        # product with X~^T for the initial residual (the reference's RHS Tdot
        # and the Tdot of A x0 go through a single pass by linearity,
        # cg_sampler.hip TD_RESID: one Tdot less than the reference's count for
        # a warm start); the linear predictor of the Omega update
        after = hip.get_dot_count()
        warm = 1 if np.any(x0 != 0.) else 0
        # (iterations enqueued past the stopping one return at entry -- every
        # operator kernel reads the solve's stop flag -- and are not counted)
        assert after[0] - before[0] == n_cg + warm + 1
        assert after[1] - before[1] == n_cg + 1
        # ---- summaries after the update (chain_summary_kernel)
        summ.update(coef_d, g_b, ls_b)
        mean_a, square_a, n_avg_a = chain.get_summary()
        assert n_avg_a == n_avg + 1
        assert np.abs(mean_a - summ.mean).max() <= 1e-12 * max(
            1., np.abs(summ.mean).max())
        assert np.abs(square_a - summ.square).max() <= 1e-12 * max(
            1., np.abs(summ.square).max())
        # ---- log posterior of the new state (chain_gscale_kernel)
        coef_a, obs_a, ls_a, g_a = chain.get_state()
        assert np.array_equal(coef_a, coef_d)
        assert np.array_equal(ls_a, kept['local_scale'][0])
        lp_o = ora.logp(coef_a, g_a, obs_a)
        lp_d = float(kept['logp'][0])
        # f32 storage rounds the CENTRED entries (X - mean is not
        # f32-representable even when X is): the stored operator differs from
        # the oracle's by 6e-8 relative per entry
        lp_tol = 1e-7 if storage == 'float32' else 1e-10
        assert abs(lp_d - lp_o) <= lp_tol * abs(lp_o), (lp_d, lp_o)
        ll_d, lp_d2 = chain.logp()
        assert lp_d2 == lp_d
        from oracle.gibbs import loglik
        ll_o = loglik(family, ora.design, ora.outcome, coef_a, obs_a)
        assert abs(ll_d - ll_o) <= lp_tol * abs(ll_o)
        assert g_a == kept['global_scale'][0] and g_a > 0
        assert np.all(ls_a > 0) and np.all(np.isfinite(ls_a))
        if family == 'logit':
            assert np.all(obs_a > 0) and obs_a.shape == (n,)
        else:
            assert obs_a > 0
    chain.close()


def test_gscale_update_modes_follow_the_reference():
    """SamplerOptions.gscale_update = 'optimize' / None on the device chain
    (bayesbridge.py:412-456): deterministic, so compared exactly."""
    from bayesbridge_amd.device_chain import HipGibbsChain
    X, (n_success, n_trial) = _problem('logit', 'sparse', n=1500, p=80)
    hip = _designs(X, 'sparse', 'auto')
    P = hip.shape[1]
    chain = HipGibbsChain(hip, 'logit', n_success, n_trial=n_trial,
                          bridge_exponent=ALPHA, slab_size=SLAB, seed=3)
    chain.set_state(np.zeros(P), None, np.ones(P - 1), .05)
    chain.init_obs_prec()
    chain.set_gscale_update(None)
    kept, _ = chain.run(3)
    assert np.all(kept['global_scale'] == .05)
    chain.set_gscale_update('optimize')
    kept, _ = chain.run(3)
    lower = .001 / unit_bridge_magnitude(ALPHA)
    for s in range(3):
        beta = kept['coef'][s][1:]
        phi = beta.size / ALPHA / np.sum(np.abs(beta) ** ALPHA)
        want = max(phi ** -(1 / ALPHA), lower)
        assert abs(kept['global_scale'][s] - want) <= 1e-12 * want
    chain.set_gscale_update('sample')
    kept, _ = chain.run(20)
    assert len(np.unique(kept['global_scale'])) == 20
    with pytest.raises(KeyError):
        chain.set_gscale_update('bogus')
    # the driver passes the option through (it was ignored)
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    prior = RegressionCoefPrior(bridge_exponent=ALPHA,
                                regularizing_slab_size=SLAB)
    unit = unit_bridge_magnitude(ALPHA)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s, _ = BayesBridge(RegressionModel((n_success, n_trial), hip, 'logit'),
                           prior).gibbs(
            4, init={'global_scale': .05, 'coef': np.zeros(P)}, seed=1,
            options={'global_scale_update': None})
    assert np.allclose(s['global_scale'], .05, rtol=1e-14)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s, _ = BayesBridge(RegressionModel((n_success, n_trial), hip, 'logit'),
                           prior).gibbs(
            4, init={'global_scale': .05, 'coef': np.zeros(P)}, seed=1,
            options={'global_scale_update': 'optimize'})
    for k in range(4):
        beta = s['coef'][1:, k]
        phi = beta.size / ALPHA / np.sum(np.abs(beta) ** ALPHA)
        want = max(phi ** -(1 / ALPHA), lower) * unit
        assert abs(s['global_scale'][k] - want) <= 1e-12 * want


def test_resume_keeps_its_own_philox_key():
    """gibbs(seed=1) -> info1; gibbs(seed=2); gibbs_resume(info1) continues
    run 1's streams (it silently continued with seed 2's)."""
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    X, y = _problem('logit', 'sparse', n=1500, p=80)
    prior = RegressionCoefPrior(bridge_exponent=ALPHA,
                                regularizing_slab_size=SLAB)
    init = {'global_scale': .05, 'coef': np.zeros(X.shape[1] + 1)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full, _ = BayesBridge(RegressionModel(y, X, 'logit'), prior).gibbs(
            8, init=dict(init), seed=1)
        b = BayesBridge(RegressionModel(y, X, 'logit'), prior)
        first, info1 = b.gibbs(4, init=dict(init), seed=1)
        b.gibbs(3, init=dict(init), seed=2)            # interleaved other run
        merged, _ = b.gibbs_resume(info1, 4, merge=True, prev_samples=first)
    assert np.array_equal(merged['coef'], full['coef'])


def _normals(seed, stream, n):
    import ctypes
    from bayesbridge_amd import _lib
    out = np.empty(n)
    _lib.check(_lib.load().bbx_device_normal(
        0, seed, stream, n, out.ctypes.data_as(ctypes.c_void_p)))
    return out


def test_device_normals_distribution_and_independence():
    """fill_normal_kernel (the chain's eta1/eta2, cg_sampler.py:61-62 uses
    np.random.randn): moments, kurtosis, KS on 1e6 draws; the two streams of
    one iteration and the streams of consecutive iterations are uncorrelated
    and share no values."""
    from bayesbridge_amd import _lib
    n = 1000000
    a = _normals(12345, _lib.STREAM_ETA1, n)
    assert np.all(np.isfinite(a))
    se = 1 / np.sqrt(n)
    assert abs(a.mean()) < 5 * se
    assert abs(a.var() - 1.) < 5 * np.sqrt(2.) * se
    assert abs(stats.skew(a)) < 5 * np.sqrt(6.) * se
    assert abs(stats.kurtosis(a)) < 5 * np.sqrt(24.) * se
    assert stats.kstest(a[:200000], 'norm').pvalue > 1e-3
    assert a.max() > 4. and a.min() < -4.          # tails are populated
    # lag-1 autocorrelation inside a stream
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5 * se
    b = _normals(12345, _lib.STREAM_ETA2, n)
    c = _normals(12346, _lib.STREAM_ETA1, n)
    for other in (b, c):
        assert abs(np.corrcoef(a, other)[0, 1]) < 5 * se
        assert abs(np.corrcoef(a ** 2, other ** 2)[0, 1]) < 5 * se
        assert np.intersect1d(a[:100000], other[:100000]).size == 0
    assert np.array_equal(a, _normals(12345, _lib.STREAM_ETA1, n))

    # the chain's per-iteration normals: what bbx_chain_eta regenerates is
    # what consecutive iterations draw -- distinct, uncorrelated streams
    from bayesbridge_amd.device_chain import HipGibbsChain
    X, (n_success, n_trial) = _problem('logit', 'sparse', n=20000, p=300)
    hip = _designs(X, 'sparse', 'auto')
    chain = HipGibbsChain(hip, 'logit', n_success, n_trial=n_trial, seed=9)
    e1a, e2a = chain.eta(0)
    e1b, e2b = chain.eta(1)
    se = 1 / np.sqrt(len(e1a))
    assert abs(np.corrcoef(e1a, e1b)[0, 1]) < 5 * se
    assert abs(np.corrcoef(e1a[:len(e2a)], e2a)[0, 1]) < 5 / np.sqrt(len(e2a))
    assert np.intersect1d(e1a, e1b).size == 0
    assert abs(e1a.var() - 1.) < 5 * np.sqrt(2.) * se
    assert abs(e2b.var() - 1.) < 6 * np.sqrt(2. / len(e2b))
    chain.seed = 10
    assert not np.array_equal(chain.eta(0)[0], e1a)
