"""GPU: the device-resident chain (Philox draws) and its scalar samplers.
Distribution parity with the reference, checked against closed forms and
against the host samplers that are themselves bit-pinned to the reference
(tests/test_oracle_vs_reference.py)."""
import ctypes
import warnings

import numpy as np
import pytest
import scipy.sparse as sparse
from scipy import stats

pytestmark = pytest.mark.gpu


def _dev_pg(seed, shape, tilt):
    from bayesbridge_amd import _lib
    lib = _lib.load()
    out = np.empty(len(tilt))
    shape = np.ascontiguousarray(shape, dtype=np.int32)
    tilt = np.ascontiguousarray(tilt, dtype=np.float64)
    _lib.check(lib.bbx_device_polya_gamma(
        0, seed, len(tilt), shape.ctypes.data_as(ctypes.c_void_p),
        tilt.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p)))
    return out


def _dev_ts(seed, a, tilt):
    from bayesbridge_amd import _lib
    lib = _lib.load()
    out = np.empty(len(tilt))
    tilt = np.ascontiguousarray(tilt, dtype=np.float64)
    _lib.check(lib.bbx_device_tilted_stable(
        0, seed, len(tilt), float(a), tilt.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p)))
    return out


def test_device_polya_gamma_moments_and_ks():
    n = 200000
    for b, c in [(1, 0.), (1, 1.5), (3, -4.), (1, 25.)]:
        x = _dev_pg(11, np.full(n, b), np.full(n, c))
        assert np.all(x > 0)
        if c == 0.:
            mean, var = b / 4., b / 24.
        else:                                  # Polson, Scott & Windle (2013)
            mean = b / (2 * c) * np.tanh(c / 2)
            var = b * (np.sinh(c) - c) / (4 * c ** 3 * np.cosh(c / 2) ** 2)
        assert abs(x.mean() - mean) < 6 * np.sqrt(var / n)
        assert abs(x.var() - var) < .03 * var
    # two-sample KS against the host sampler on the reference's stream
    from bayesbridge_amd.hostrng import ReferenceRandom
    host = ReferenceRandom(3)
    tilt = np.full(50000, 1.2)
    a = _dev_pg(5, np.ones(50000, dtype=np.int32), tilt)
    b_ = host.polya_gamma(np.ones(50000, dtype=np.int32), tilt)
    assert stats.ks_2samp(a, b_).pvalue > 1e-3
    # same seed => same draws; other seed => other draws
    assert np.array_equal(a, _dev_pg(5, np.ones(50000, dtype=np.int32), tilt))
    assert not np.array_equal(a, _dev_pg(6, np.ones(50000, dtype=np.int32),
                                         tilt))


def test_device_polya_gamma_draws_do_not_depend_on_the_elements_per_lane():
    """csrc/pg_queue.hpp: a lane takes E elements (1, 4 or 8 by vector length),
    every piece of a draw has its own Philox sub-stream of the element -- so
    the draw of element i depends on (seed, i, shape, tilt) only: a long vector
    (8 per lane) starts with the draws of its short prefix (1 per lane), bit
    for bit, binomial shapes (the sequential sampler) mixed in.  And across the
    range of tilts -- the mixture weight formed from products below |psi| = 40
    (samplers.hpp right_mass_direct), from logarithms beyond -- the draws pass
    a two-sample KS test against the host sampler on the reference's stream
    (random/polya_gamma/polya_gamma.pyx:40-74)."""
    rng = np.random.default_rng(4)
    n_long, n_short = 120000, 30000
    shape = np.ones(n_long, dtype=np.int32)
    shape[::7] = rng.integers(2, 6, size=len(shape[::7]))
    tilt = rng.normal(0., 3., n_long)
    tilt[::11] = rng.normal(0., 30., len(tilt[::11]))      # |psi| up to ~100
    long_ = _dev_pg(9, shape, tilt)
    short = _dev_pg(9, shape[:n_short], tilt[:n_short])
    assert np.array_equal(long_[:n_short], short)
    assert np.all(np.isfinite(long_)) and np.all(long_ > 0)
    # a non-finite tilt comes back as NaN (no rejection loop spins on it)
    bad = tilt[:4096].copy()
    bad[[5, 77, 900]] = [np.nan, np.inf, -np.inf]
    out = _dev_pg(9, shape[:4096], bad)
    assert np.all(np.isnan(out[[5, 77, 900]]))
    keep = np.ones(4096, dtype=bool)
    keep[[5, 77, 900]] = False
    assert np.array_equal(out[keep], long_[:4096][keep])
    from bayesbridge_amd.hostrng import ReferenceRandom
    host = ReferenceRandom(8)
    for c in (.05, .7, 3., 8., 38., 45., 120.):
        ones, t = np.ones(60000, dtype=np.int32), np.full(60000, c)
        a, b_ = _dev_pg(21, ones, t), host.polya_gamma(ones, t)
        assert stats.ks_2samp(a, b_).pvalue > 1e-3, c
        mean = 1. / (2 * c) * np.tanh(c / 2)
        assert abs(a.mean() - mean) < 6 * a.std() / np.sqrt(len(a)), c


def test_device_tilted_stable_laplace_transform_and_ks():
    # X ~ exp(-lam x) f_a(x) / E, f_a positive stable with E exp(-s S) =
    # exp(-s^a):  E exp(-s X) = exp(-((s + lam)^a - lam^a))
    n = 200000
    # tilt^a = .84, 2.3, 2, 1.9 and -- deep in the double-rejection regime,
    # whose inner loop the device flattens (samplers.hpp dr_trial_flat) --
    # 4.2, 8.4, 20
    for a, lam in [(.25, .5), (.25, 30.), (.5, 4.), (.125, 200.),
                   (.25, 300.), (.25, 5000.), (.5, 400.)]:
        x = _dev_ts(9, a, np.full(n, lam))
        assert np.all(x > 0) and np.all(np.isfinite(x))
        for s in (.3, 2.):
            want = np.exp(-((s + lam) ** a - lam ** a))
            got = np.exp(-s * x)
            assert abs(got.mean() - want) < 6 * got.std() / np.sqrt(n)
    from bayesbridge_amd.hostrng import ReferenceRandom
    host = ReferenceRandom(4)
    for lam in (3., 100., 2000.):        # tilt^a = 1.3 (plain), 3.2, 6.7
        tl = np.full(50000, lam)
        assert stats.ks_2samp(_dev_ts(2, .25, tl),
                              host.tilted_stable(.25, tl)).pvalue > 1e-3, lam


def test_device_tilted_stable_mixed_tilts_match_the_host_sampler():
    """One launch with tilts spread over both regimes (what the chain's
    local-scale update looks like: plain rejection and the flattened double
    rejection side by side in every block), compared bin by bin with the host
    sampler that is bit-pinned to the reference."""
    from bayesbridge_amd.hostrng import ReferenceRandom
    rng = np.random.default_rng(0)
    n, a = 240000, .25
    tp = np.concatenate((rng.uniform(.05, 9., n // 2),
                         rng.uniform(9., 150., n // 2)))     # tilt^a
    tilt = tp ** (1 / a)
    dev = _dev_ts(5, a, tilt)
    host = ReferenceRandom(6).tilted_stable(a, tilt)
    assert np.all(dev > 0) and np.all(np.isfinite(dev))
    scale = tilt ** .5                   # brings the bins to comparable ranges
    for lo, hi in [(0, 1), (1, 2), (2, 2.5), (2.5, 3), (3, 4), (4, 6), (6, 9),
                   (9, 40), (40, 150)]:
        m = (tp >= lo) & (tp < hi)
        x, y = scale[m] / np.sqrt(dev[m]), scale[m] / np.sqrt(host[m])
        assert stats.ks_2samp(x, y).pvalue > 1e-3, (lo, hi)
        assert abs(np.log(x).mean() - np.log(y).mean()) < \
            5 * np.log(y).std() * np.sqrt(2. / m.sum()), (lo, hi)


def test_device_gamma_moments():
    from bayesbridge_amd import _lib
    lib = _lib.load()
    for shape in (.4, 3., 2.5e5):
        out = np.empty(100000)
        _lib.check(lib.bbx_device_gamma(0, 7, out.size, shape,
                                        out.ctypes.data_as(ctypes.c_void_p)))
        assert abs(out.mean() - shape) < 6 * np.sqrt(shape / out.size)
        assert abs(out.var() - shape) < .05 * shape


def _problem(n=3000, p=200, seed=3, model='logit'):
    from bayesbridge_amd import simulate
    X = simulate.simulate_design_csr(n, p, binary_frac=.8,
                                     binary_pred_freq=.1, seed=seed)
    beta = np.zeros(p)
    beta[:5], beta[5:10] = 1.5, -1.
    y = simulate.simulate_outcome(X, beta, model, seed=seed + 1)
    return X, y, beta


@pytest.mark.parametrize("model", ['logit', 'linear'])
def test_device_chain_is_reproducible_and_resumable(model):
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    X, y, _ = _problem(model=model)
    prior = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    init = {'global_scale': .05, 'coef': np.zeros(X.shape[1] + 1)}

    def run(n_iter, seed):
        b = BayesBridge(RegressionModel(y, X, model), prior)
        return b, b.gibbs(n_iter, init=dict(init), seed=seed,
                          params_to_save='all')
    _, (s1, i1) = run(8, 21)
    _, (s2, _) = run(8, 21)
    _, (s3, _) = run(8, 22)
    assert np.array_equal(s1['coef'], s2['coef'])           # bitwise
    assert not np.array_equal(s1['coef'], s3['coef'])
    assert np.all(np.isfinite(s1['coef'])) and np.all(np.isfinite(s1['logp']))
    assert s1['coef'].shape == (X.shape[1] + 1, 8)
    assert s1['local_scale'].shape == (X.shape[1], 8)
    assert np.all(i1['_reg_coef_sampling_info']['n_cg_iter'] > 0)
    # 4 + resume 4 == 8 straight (Philox counters are keyed by iteration)
    b, (sa, ia) = run(4, 21)
    sb, ib = BayesBridge(RegressionModel(y, X, model), prior).gibbs_resume(
        ia, 4, merge=True, prev_samples=sa)
    assert np.array_equal(sb['coef'], s1['coef'])
    assert np.allclose(sb['global_scale'], s1['global_scale'], rtol=1e-12)


def test_a_chain_run_straight_after_creation_starts_from_a_defined_state():
    """bbx_chain_create leaves every state vector defined: coef = 0, unit
    scales and -- logit -- Omega = the Polya-Gamma mean at psi = 0
    (logistic_model.py:80-87), so a chain that is run without set_state /
    init_obs_prec does not read whatever the allocator returned (a recycled
    buffer full of NaN gave "non-finite residual inside CG").  Two chains
    created around a poisoned allocation draw the same, finite samples."""
    import torch
    from bayesbridge_amd import (HipGibbsChain, HipSparseDesignMatrix,
                                 simulate)
    n, p = 60000, 400
    X = simulate.simulate_binary_csr_fast(n, p, .03, seed=4)
    y = (np.random.default_rng(1).random(n) < .3).astype(np.float64)
    hip = HipSparseDesignMatrix(X, center_predictor=False, add_intercept=True)
    outs = []
    for _ in range(2):
        # recycle device memory holding NaN through the allocators
        junk = [torch.full((n,), float('nan'), dtype=torch.float64,
                           device='cuda') for _ in range(6)]
        del junk
        torch.cuda.empty_cache()
        chain = HipGibbsChain(hip, 'logit', y, sd_unshrunk=[2.], slab_size=2.,
                              seed=7)
        coef, obs, ls, g = chain.get_state()
        assert np.all(coef == 0.) and np.all(ls == 1.) and g == 1.
        assert np.all(obs == .5)          # n_trial / 2 at psi = 0, as the reference
        out, n_bad = chain.run(3)
        assert n_bad == 0 and np.all(np.isfinite(out['coef']))
        outs.append(out['coef'])
        chain.close()
    assert np.array_equal(outs[0], outs[1])


def test_device_chain_agrees_with_reference_stream_chain():
    """Same posterior, different random streams: posterior means of the large
    coefficients and of log tau agree within Monte Carlo error."""
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    X, y, beta = _problem(n=2000, p=60, seed=5)
    prior = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    init = {'global_scale': .05, 'coef': np.zeros(X.shape[1] + 1)}
    n_iter, burn = 600, 150
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sd, _ = BayesBridge(RegressionModel(y, X, 'logit'), prior).gibbs(
            n_iter, n_burnin=burn, init=dict(init), seed=1)
        sr, _ = BayesBridge(RegressionModel(y, X, 'logit'), prior).gibbs(
            n_iter, n_burnin=burn, init=dict(init), seed=1,
            options={'rng': 'reference'})
    md, mr = sd['coef'].mean(axis=1), sr['coef'].mean(axis=1)
    sdv = sr['coef'].std(axis=1)
    big = np.abs(mr) > .3
    assert big.sum() >= 5
    # effective sample size is well below n_iter - burn: allow 0.5 posterior sd
    assert np.all(np.abs(md - mr)[big] < .5 * sdv[big] + .02)
    lg_d, lg_r = np.log(sd['global_scale']), np.log(sr['global_scale'])
    assert abs(lg_d.mean() - lg_r.mean()) < .5 * lg_r.std() + .1


def test_dense_linear_chain_config1_summary(golden_dir):
    """BASELINE config 1 (dense 2000x500 linear, cg): the reference's 20
    iterations are summarised in the fixture.  The exact-seed mode reproduces
    them at the reference's own CPU-vs-GPU bound (atol 1e-5 on the samples,
    tests/gpu_tests/test_gibbs.py:44) -- the CPU oracle's own chain moves by
    <= 1.2e-6 and its stopping iterations by <= 4 when X is perturbed by 1e-15
    (profiles/r06_config1_sensitivity.txt) -- and the device-RNG mode, other
    random streams, climbs the same transient: tau from .005 to ~.02 within
    the 20 iterations, the 15 signals found, CG effort growing alike."""
    import os
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel, simulate
    g = np.load(os.path.join(golden_dir,
                             'chain_linear_dense_2000x500_summary.npz'))
    np.random.seed(111)
    X = np.random.randn(2000, 500)    # simulate_design(..., 'dense', seed=111)
    assert np.allclose(X[:4, :4], g['X_head'])
    beta = simulate.demo_beta(500)
    y = simulate.simulate_outcome(X, beta, 'linear', seed=1)
    assert np.allclose(y[:8], g['y_head'])
    prior = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s, info = BayesBridge(RegressionModel(y, X, 'linear'), prior).gibbs(
            20, 0, init={'global_scale': .01}, coef_sampler_type='cg',
            seed=111, options={'rng': 'reference'})
    assert np.allclose(s['coef'][:, -1], g['coef_last'], atol=1e-5)
    assert np.allclose(s['coef'][:, 10:].mean(axis=1), g['coef_mean_last10'],
                       atol=1e-5)
    assert np.allclose(s['global_scale'], g['global_scale'], rtol=1e-5)
    assert np.allclose(s['logp'], g['logp'], rtol=1e-6)
    # ~80 CG iterations on a flat stretch of the residual curve: the oracle's
    # own stopping iteration moves by up to 4 under 1e-15 perturbations
    assert np.abs(info['_reg_coef_sampling_info']['n_cg_iter']
                  - g['n_cg_iter']).max() <= 5

    # the device-RNG mode (what gibbs() runs by default) on the same problem
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d, dinfo = BayesBridge(RegressionModel(y, X, 'linear'), prior).gibbs(
            20, 0, init={'global_scale': .01}, coef_sampler_type='cg',
            seed=111)
    assert dinfo['options']['rng'] == 'device'
    m_d, m_r = d['coef'][:, 10:].mean(axis=1), g['coef_mean_last10']
    # posterior sd of a coefficient ~ 1 / sqrt(n) = .022; ten correlated draws
    assert np.abs(m_d - m_r)[1:16].max() < .25   # (still in the transient)
    assert np.abs(m_d[16:]).max() < .12 and np.abs(m_r[16:]).max() < .12
    assert np.all((d['global_scale'][:4] > .003) & (d['global_scale'][:4] < .008))
    assert np.all((d['global_scale'][14:] > .012) & (d['global_scale'][14:] < .04))
    n_d = dinfo['_reg_coef_sampling_info']['n_cg_iter']
    assert n_d[:4].max() <= 12 and 50 <= n_d[15:].mean() <= 110
    assert abs(d['logp'][10:].mean() - g['logp'][10:].mean()) \
        < 3 * max(g['logp'][10:].std(), d['logp'][10:].std())


def _variant_chain(tmp_path, env, family='logit', iters=6, n=4000, p=300):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    tag = "_".join("%s%s" % kv for kv in sorted(env.items()))
    out = os.path.join(str(tmp_path), "chain_%s_%s.npz" % (family, tag))
    run = subprocess.run(
        [sys.executable, os.path.join(ROOT, "scripts", "chain_variant_run.py"),
         out, family, str(n), str(p), str(iters)],
        env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    return np.load(out)


@pytest.mark.parametrize("family", ['logit', 'linear'])
def test_two_stream_iteration_is_bitwise_the_one_stream_iteration(tmp_path, family):
    """chain_step runs the Omega update and the tau / lambda updates on two
    streams for designs from BASELINE config 2's size on (BBX_CHAIN_FORK unset:
    n >= 50 000 and >= 2 048 shrunk coefficients).  Forced on and off on a
    small problem: every saved sample is bit-identical (the
    branches share no data, Philox streams are keyed by element)."""
    a = _variant_chain(tmp_path, {'BBX_CHAIN_FORK': '0'}, family)
    b = _variant_chain(tmp_path, {'BBX_CHAIN_FORK': '1'}, family)
    for key in ('coef', 'local_scale', 'obs_prec', 'global_scale', 'logp',
                'n_cg_iter'):
        assert np.array_equal(a[key], b[key]), key
    assert np.all(np.isfinite(a['logp'])) and a['coef'].shape[0] == 6


def test_automatic_two_stream_choice_matches_one_stream(tmp_path):
    """BBX_CHAIN_FORK unset on a design past the automatic threshold (60 000
    rows, 2 500 shrunk coefficients): the chain forks by itself, and its
    samples are bit for bit those of the forced one-stream run."""
    auto = _variant_chain(tmp_path, {}, n=60000, p=2500, iters=4)
    one = _variant_chain(tmp_path, {'BBX_CHAIN_FORK': '0'}, n=60000, p=2500,
                         iters=4)
    for key in ('coef', 'local_scale', 'obs_prec', 'global_scale', 'logp',
                'n_cg_iter'):
        assert np.array_equal(auto[key], one[key]), key


def test_status_updates_are_printed_in_the_device_mode_too(capsys):
    """gibbs(n_status_update=k) prints the reference's k status lines
    (gibbs_util.py:214-238) in the default device-RNG mode as well -- from the
    library's host loop through bbx_chain_set_progress -- and printing changes
    no sample."""
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    X, y, _ = _problem(n=1500, p=40, seed=2)
    prior = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    init = {'global_scale': .05, 'coef': np.zeros(X.shape[1] + 1)}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bridge = BayesBridge(RegressionModel(y, X, 'logit'), prior)
        capsys.readouterr()
        quiet, _ = bridge.gibbs(12, init=dict(init), seed=4)
        assert "iterations complete" not in capsys.readouterr().out
        loud, _ = bridge.gibbs(12, init=dict(init), seed=4, n_status_update=3)
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if "Gibbs iterations complete" in ln]
    assert [int(ln.split()[0]) for ln in lines] == [4, 8, 12], out
    assert all("has elasped since the last update." in ln for ln in lines)
    assert np.array_equal(quiet['coef'], loud['coef'])
    # more updates than iterations: one line per iteration, as the reference
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bridge.gibbs(3, init=dict(init), seed=4, n_status_update=10)
    assert capsys.readouterr().out.count("iterations complete") == 3
