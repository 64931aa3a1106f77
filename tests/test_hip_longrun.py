"""GPU: long-run DISTRIBUTION parity of the chain that bench.py times and
BayesBridge.gibbs runs by default -- every draw from the device's Philox
streams (options['rng'] = 'device').

Those streams cannot be compared draw by draw with the reference loop
(bayesbridge.py:210-240: MT19937 normals, PCG64 Polya-Gamma / tilted-stable
draws), and the reference's own GPU contract (tests/gpu_tests/test_gibbs.py:
34-44, atol=1e-5 on samples) presumes shared streams.  The statistical
counterpart: 30 000 post-burn-in device iterations per problem against a
fixture of 4 x 25 000 iterations of the IMPORTED REFERENCE
(tests/golden/make_longrun.py; numbers only), |z| < 4.5 for the ergodic mean
AND variance of every coefficient, of log tau, of the log posterior, of every
log lambda_j and of the mean Omega, with batch-means Monte-Carlo standard
errors of both sides.  A seeded negative control (every Polya-Gamma draw
scaled by 1.05 before the next coefficient draw reads it) must fail the same
comparison, and the random streams of one iteration must be uncorrelated
(Omega draws against eta1, lambda draws against eta2).
"""
import os
import warnings

import numpy as np
import pytest

import longrun_cases as lc

pytestmark = pytest.mark.gpu

CHUNK = 2500             # iterations per gibbs / gibbs_resume call


def _fixture(golden_dir, name, case):
    g = np.load(os.path.join(golden_dir, 'longrun_%s.npz' % name))
    assert np.allclose(lc.case_checksum(case), g['checksum'], rtol=1e-12), \
        "regenerated problem is not the fixture's"
    assert list(g['names']) == lc.series_names(case)
    assert int(g['batch']) == lc.BATCH
    return {k: g[k] for k in ('mean', 'mean_se', 'var', 'var_se')}


def _bridge(case):
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    old = {k: os.environ.get(k) for k in case['env']}
    os.environ.update(case['env'])
    try:
        X = case['X'].copy()
        model = RegressionModel(case['outcome'], X, case['family'])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return BayesBridge(model, RegressionCoefPrior(**case['prior_kw']))


def _device_series(case, seed, keep=lc.DEV_KEEP, omega_scale=None,
                   lambda_scale=None):
    """Per-iteration statistics [keep, K] of one device-RNG chain through
    BayesBridge.gibbs / gibbs_resume (the default mode), in chunks so that
    the n x T array of Omega samples never exists at once.  omega_scale: the
    negative control -- the chain is stepped one iteration at a time and
    Omega (the scalar noise precision for the linear model) is multiplied by
    it between the Polya-Gamma draw and the next coefficient draw
    (lambda_scale: the local scales likewise)."""
    if lambda_scale is not None and omega_scale is None:
        omega_scale = 1.
    bridge = _bridge(case)
    parts, n_cg = [], []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if omega_scale is None:
            s, info = bridge.gibbs(
                lc.BURNIN + CHUNK, n_burnin=lc.BURNIN, seed=seed,
                init=dict(case['init']), params_to_save='all',
                coef_sampler_type='cg')
            assert info['options']['rng'] == 'device'
            while True:
                parts.append(lc.series(case, s))
                n_cg.append(info['_reg_coef_sampling_info']['n_cg_iter'])
                if sum(len(p_) for p_ in parts) >= keep:
                    break
                s, info = bridge.gibbs_resume(info, CHUNK)
        else:
            bridge.gibbs(lc.BURNIN, n_burnin=lc.BURNIN, seed=seed,
                         init=dict(case['init']), coef_sampler_type='cg')
            chain = bridge._chain
            rows = {k: [] for k in ('coef', 'local_scale', 'obs_prec',
                                    'global_scale', 'logp')}
            for _ in range(keep):
                coef, obs, ls, _ = chain.get_state()
                chain.set_state(obs_prec=np.asarray(obs) * omega_scale)
                if lambda_scale is not None:
                    chain.set_state(local_scale=ls * lambda_scale)
                out, _ = chain.run(1, save=('coef', 'local_scale',
                                            'obs_prec'))
                for k in rows:
                    rows[k].append(out[k][0])
            s = {'coef': np.ascontiguousarray(np.array(rows['coef']).T),
                 'local_scale': np.ascontiguousarray(
                     np.array(rows['local_scale']).T),
                 'global_scale': np.array(rows['global_scale']),
                 'logp': np.array(rows['logp']),
                 # (C-contiguous like the chunked path's array: NumPy's
                 # reductions add in another order over a transposed view)
                 'obs_prec': np.ascontiguousarray(np.array(rows['obs_prec']).T)
                 if case['family'] == 'logit'
                 else np.array(rows['obs_prec'])[:, 0]}
            bridge.prior.adjust_scale(s['global_scale'], s['local_scale'],
                                      to='coef_magnitude')
            parts.append(lc.series(case, s))
    S = np.concatenate(parts)[:keep]
    return S, bridge, (np.concatenate(n_cg) if n_cg else None)


def _compare(name, dev, ref, names):
    zm, zv = lc.z_scores(dev, ref)
    worst_m, worst_v = np.abs(zm).argmax(), np.abs(zv).argmax()
    report = ("%s: max |z| mean %.2f (%s), variance %.2f (%s); rms z %.2f / "
              "%.2f over %d statistics"
              % (name, abs(zm[worst_m]), names[worst_m], abs(zv[worst_v]),
                 names[worst_v], np.sqrt((zm ** 2).mean()),
                 np.sqrt((zv ** 2).mean()), len(zm)))
    return zm, zv, report


@pytest.mark.parametrize("name", lc.CASES)
def test_device_rng_chain_matches_reference_long_run(golden_dir, name):
    """30 000 kept iterations of the default (device-RNG) chain against the
    reference's 4 x 25 000: every ergodic mean and variance within 4.5
    combined Monte-Carlo standard errors; the z scores as a whole look like
    standard normals (rms below 1.5 -- a 10 % mis-scaled draw anywhere moves
    dozens of them)."""
    case = lc.make_case(name)
    ref = _fixture(golden_dir, name, case)
    S, bridge, n_cg = _device_series(case, seed=20261)
    if name == 'logit_binary_packed':
        info = bridge.model.design.tiled_info()
        assert info['X']['packed'] and info['Xt']['packed']
    dev = lc.batch_stats([S])
    names = lc.series_names(case)
    zm, zv, report = _compare(name, dev, ref, names)
    print(report)
    assert np.all(np.isfinite(zm)) and np.all(np.isfinite(zv)), report
    assert np.abs(zm).max() < lc.Z_MAX, report
    assert np.abs(zv).max() < lc.Z_MAX, report
    assert np.sqrt((zm ** 2).mean()) < 1.5, report
    assert np.sqrt((zv ** 2).mean()) < 1.5, report
    # CG effort: a property of the solver, not of the posterior -- the means
    # agree to a fraction of an iteration
    g = np.load(os.path.join(golden_dir, 'longrun_%s.npz' % name))
    assert abs(n_cg.mean() - g['mean_n_cg'].mean()) < 1., \
        (n_cg.mean(), g['mean_n_cg'])


def test_negative_control_omega_scaled_by_5_percent_fails(golden_dir):
    """The same comparison must FAIL when every Polya-Gamma draw is 5 % too
    large by the time the coefficient draw reads it (the size of error the
    short whole-chain checks of rounds 1-5 could not see): dozens of
    statistics land beyond the bound."""
    name = 'logit_mixed_ntrial'
    case = lc.make_case(name)
    ref = _fixture(golden_dir, name, case)
    # (12 000 iterations: the error shows at |z| > 100, a third of the length of
    # the real comparison is plenty)
    S, _, _ = _device_series(case, seed=20261, keep=12000, omega_scale=1.05)
    dev = lc.batch_stats([S])
    zm, zv, report = _compare(name + ' [Omega x 1.05]', dev, ref,
                              lc.series_names(case))
    print(report)
    assert np.abs(zm).max() > 2 * lc.Z_MAX, report
    assert (np.abs(zm) > lc.Z_MAX).sum() >= 5, report
    assert np.sqrt((zm ** 2).mean()) > 1.5, report


def test_negative_control_local_scales_scaled_by_5_percent_fails(golden_dir):
    """... and likewise when every lambda_j is 5 % too large by the time the
    coefficient draw reads it (8 000 iterations suffice: the coefficients are
    shrunk less, the full-length run shows |z| = 33 and 98 statistics beyond
    the bound, profiles/r06_longrun_power.txt)."""
    name = 'logit_mixed_ntrial'
    case = lc.make_case(name)
    ref = _fixture(golden_dir, name, case)
    S, _, _ = _device_series(case, seed=20261, keep=8000, lambda_scale=1.05)
    zm, zv, report = _compare(name + ' [lambda x 1.05]', lc.batch_stats([S]),
                              ref, lc.series_names(case))
    print(report)
    assert max(np.abs(zm).max(), np.abs(zv).max()) > 2 * lc.Z_MAX, report
    assert (np.abs(zm) > lc.Z_MAX).sum() + (np.abs(zv) > lc.Z_MAX).sum() >= 5, \
        report


def test_negative_control_passes_unscaled(golden_dir):
    """... and the harness of the control itself is sound: stepped one
    iteration at a time with the state pulled and pushed back unchanged
    (scale 1.0) the chain is bit for bit the chain of the chunked run, so the
    failure above is the 5 %, not the stepping."""
    for name in ('logit_mixed_ntrial', 'linear_dense'):
        case = lc.make_case(name)
        S1, _, _ = _device_series(case, seed=5, keep=300, omega_scale=1.0)
        S2, _, _ = _device_series(case, seed=5, keep=300)
        assert np.array_equal(S1, S2[:300]), name


def _rank(a):
    r = np.empty(len(a))
    r[np.argsort(a, kind='stable')] = np.arange(len(a))
    return (r - r.mean()) / r.std()


def _partial_out(y, x, bins=64):
    """y minus its mean inside quantile bins of x."""
    order = np.argsort(x, kind='stable')
    out = np.empty(len(y))
    for idx in np.array_split(order, bins):
        out[idx] = y[idx] - y[idx].mean()
    return out


def test_streams_of_one_iteration_are_uncorrelated():
    """Cross-stream independence inside the device chain (Philox streams
    keyed by (seed, iteration, stream id, element)): the randomness of the
    Polya-Gamma draws is uncorrelated with the eta1 that perturbed the same
    rows in the SAME iteration and with the eta1 that meets those Omega_i in
    the NEXT coefficient draw; likewise the lambda draws and eta2.  The
    legitimate dependence (Omega_i on psi_i, lambda_j on |beta_j| / tau, both
    functions of this iteration's eta) is removed first: Omega through its
    conditional mean n/(2 psi) tanh(psi / 2) (logistic_model.py:80-87),
    lambda by centring its rank inside 64 quantile bins of |beta_j| / tau
    (and |eta2_j|, which the same ratio predicts, likewise).  Bound:
    5 / sqrt(number of pooled pairs)."""
    from bayesbridge_amd import (HipGibbsChain, HipSparseDesignMatrix,
                                 simulate)
    n, p = 100000, 10000
    X = simulate.simulate_binary_csr_fast(n, p, .01, seed=31)
    beta = simulate.demo_beta(p)
    n_success, n_trial = simulate.simulate_outcome(X, beta, 'logit', seed=2)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True)
    chain = HipGibbsChain(hip, 'logit', n_success, n_trial=n_trial,
                          sd_unshrunk=[float('inf')], slab_size=2., seed=77)
    chain.set_state(global_scale=.01)
    chain.init_obs_prec()
    chain.run(20)
    pairs = {k: ([], []) for k in (
        'omega/eta1 same', 'omega/|eta1| same', 'omega/eta1 next',
        'omega/|eta1| next', '|omega|/|eta1| next', 'lambda/eta2 same',
        'lambda/|eta2| same', 'lambda/eta2 next', 'lambda/|eta2| next',
        '|lambda|/|eta2| next')}

    def add(key, f, res):
        pairs[key][0].append(f - f.mean())
        pairs[key][1].append(res - res.mean())

    for _ in range(6):
        it = chain.iteration
        chain.run(1)
        coef, om, ls, tau = chain.get_state()
        psi = hip.dot(coef)
        half = np.where(np.abs(psi) > 1e-8, np.tanh(psi / 2) / (2 * psi), .25)
        r_om = om - n_trial * half
        ratio = np.abs(coef[1:]) / tau
        r_ls = _partial_out(_rank(ls), ratio)
        a1, a2 = chain.eta(it)
        b1, b2 = chain.eta(it + 1)
        a2, b2 = a2[1:], b2[1:]
        add('omega/eta1 same', a1, r_om)
        add('omega/|eta1| same', np.abs(a1), r_om)
        add('omega/eta1 next', b1, r_om)
        add('omega/|eta1| next', np.abs(b1), r_om)
        add('|omega|/|eta1| next', np.abs(b1), np.abs(r_om))
        add('lambda/eta2 same', a2, r_ls)
        add('lambda/|eta2| same', _partial_out(np.abs(a2), ratio), r_ls)
        add('lambda/eta2 next', b2, r_ls)
        add('lambda/|eta2| next', np.abs(b2), r_ls)
        add('|lambda|/|eta2| next', np.abs(b2), np.abs(r_ls))
    for key, (f, res) in pairs.items():
        f, res = np.concatenate(f), np.concatenate(res)
        corr = np.corrcoef(f, res)[0, 1]
        assert abs(corr) < 5. / np.sqrt(len(f)), (key, corr, len(f))
