"""Build-container only (needs /root/reference): the oracle, the HIP-free host
samplers of the product and the synthetic generator against the imported
reference itself."""
import ctypes

import numpy as np
import pytest
import scipy.sparse as sparse

import oracle

pytestmark = pytest.mark.needs_reference


@pytest.fixture(scope="module")
def ref():
    import ref_import
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return ref_import.import_reference()


def test_operator_and_sampler_equal_reference(ref):
    import warnings
    warnings.simplefilter("ignore")
    bb, refsim = ref
    from bayesbridge.design_matrix import SparseDesignMatrix
    from bayesbridge.reg_coef_sampler.cg_sampler import \
        ConjugateGradientSampler
    np.random.seed(3)
    X = refsim.simulate_design(300, 40, binary_frac=.5, format_='sparse')
    rd = SparseDesignMatrix(X, center_predictor=True, add_intercept=True)
    od = oracle.OracleSparseDesign(X, center_predictor=True,
                                   add_intercept=True)
    n, P = rd.shape
    v, w = np.random.randn(P), np.random.randn(n)
    assert np.array_equal(rd.dot(v), od.dot(v))
    assert np.array_equal(rd.Tdot(w), od.Tdot(w))
    omega = np.random.gamma(2, 1, n)
    phi = np.concatenate(([0.], 1 / np.random.gamma(1, .3, P - 1)))
    z, x0, sd = np.random.randn(P), .1 * np.random.randn(P), 1.3 * np.ones(P)
    np.random.seed(7)
    c_ref, i_ref = ConjugateGradientSampler(1).sample(
        rd, omega, phi, z, coef_cg_init=x0, precond_by='prior',
        coef_scaled_sd=sd, maxiter=500, atol=1e-5 * np.sqrt(P))
    np.random.seed(7)
    e1, e2 = np.random.randn(n), np.random.randn(P)
    c_o, i_o = oracle.cg_sample(od, omega, phi, z, x0, sd, 1, e1, e2, 500,
                                1e-5 * np.sqrt(P))
    assert i_o['n_iter'] == i_ref['n_iter']
    assert np.abs(c_o - c_ref).max() <= 1e-12


def _seeds(seed):
    np.random.seed(seed)
    a = np.random.randint(1, 1 + np.iinfo(np.int32).max)
    b = np.random.randint(1, 1 + np.iinfo(np.int32).max)
    return a, b


def test_host_and_oracle_samplers_equal_reference_streams(ref):
    """Same seeds => same draws and same final PCG64 state as the reference's
    Cython samplers, for the product's libbbx_hostrng and the oracle's C."""
    from bayesbridge.random import BasicRandom
    from bayesbridge_amd.hostrng import ReferenceRandom
    from oracle.rng import OracleRandom
    rng = np.random.default_rng(0)
    n = 5000
    shape = (1 + rng.binomial(20, .3, n)).astype(np.intc)
    tilt = rng.normal(0, 3, n)
    tilt[:10], tilt[10:20], tilt[20:30] = 0., 40., 1e-8
    tl = np.exp(rng.normal(0, 4, n))
    rg = BasicRandom()
    rg.set_seed(5)
    ref_pg = rg.pg.rand_polyagamma(shape, tilt)
    for cls in (ReferenceRandom, OracleRandom):
        mine = cls(5)
        out = mine.polya_gamma(shape, tilt)
        assert np.abs(out - ref_pg).max() <= 1e-14 * ref_pg.max()
        bg = mine.pg_bitgen if hasattr(mine, 'pg_bitgen') else mine.pg
        assert bg.state['state'] == rg.pg.get_state()['state']
    for a in (.25, .125, .5, .9):
        rg.set_seed(5)
        ref_ts = rg.ts.sample(a, tl)
        for cls in (ReferenceRandom, OracleRandom):
            mine = cls(5)
            out = mine.tilted_stable(a, tl)
            assert np.array_equal(out, ref_ts)


def test_csr_generator_replays_simulate_design(ref):
    from bayesbridge_amd import simulate
    bb, refsim = ref
    for kw in (dict(binary_frac=1., binary_pred_freq=.05),
               dict(binary_frac=.5), dict(binary_frac=.9)):
        A = refsim.simulate_design(400, 50, seed=11, format_='sparse', **kw)
        B = simulate.simulate_design_csr(400, 50, seed=11, **kw)
        assert (A != B).nnz == 0


def test_prior_and_logit_validation_equal_reference(ref):
    """The product's own RegressionCoefPrior / LogisticModel checks against the
    reference's (prior.py:7-208, logistic_model.py:10-47): same attributes,
    same get_info, same hyper-parameter solution, same accept/reject."""
    import warnings
    from bayesbridge.prior import RegressionCoefPrior as RefPrior
    from bayesbridge.model.logistic_model import LogisticModel as RefLogit
    from bayesbridge_amd.model import LogisticModel
    from bayesbridge_amd.prior import RegressionCoefPrior
    cases = [
        dict(),
        dict(bridge_exponent=.25, regularizing_slab_size=1.,
             global_scale_prior_hyper_param={'log10_mean': -4., 'log10_sd': 1.}),
        dict(bridge_exponent=1., n_fixed_effect=3,
             sd_for_fixed_effect=[1., 2., 3.], sd_for_intercept=2.,
             global_scale_prior_hyper_param={'log10_mean': -2., 'log10_sd': .5},
             _global_scale_parametrization='raw'),
        dict(bridge_exponent=1 / 16, n_fixed_effect=2, sd_for_fixed_effect=4.,
             global_scale_prior_hyper_param={'log10_mean': -4.,
                                             'log10_sd': .01}),
    ]
    for kw in cases:
        mine, theirs = RegressionCoefPrior(**kw), RefPrior(**kw)
        a, b = mine.get_info(), theirs.get_info()
        assert list(a) == list(b)
        for key in a:
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key
        for hyp in ('shape', 'rate'):
            x = mine.param['gscale_neg_power'][hyp]
            y = theirs.param['gscale_neg_power'][hyp]
            assert abs(x - y) <= 1e-10 * max(abs(y), 1e-300), (kw, hyp)
        assert np.array_equal(mine.sd_for_fixed, theirs.sd_for_fixed)
        assert mine.clone(bridge_exponent=.5).get_info()['bridge_exponent'] == .5
        for to in ('raw', 'coef_magnitude'):
            g1, l1 = mine.adjust_scale(.3, np.arange(1., 5.), to)
            g2, l2 = theirs.adjust_scale(.3, np.arange(1., 5.), to)
            assert g1 == g2 and np.array_equal(l1, l2)
            ga, gb = np.array([.3, .4]), np.array([.3, .4])
            mine.adjust_scale(ga, np.ones(2), to)
            theirs.adjust_scale(gb, np.ones(2), to)
            assert np.array_equal(ga, gb)        # in place, like the reference
    for bad in (dict(bridge_exponent=2.5),
                dict(n_fixed_effect=2, sd_for_fixed_effect=[1.]),
                dict(global_scale_prior_hyper_param={'log10_mean': 0.})):
        for cls in (RegressionCoefPrior, RefPrior):
            with pytest.raises(ValueError):
                cls(**bad)

    class _D():
        shape = (4, 3)
    ok = [(np.array([0, 1, 1, 0]), None),
          (np.array([0, 2, 1, 3]), np.array([1, 2, 3, 3]))]
    bad = [(np.array([0, 2, 1, 0]), None),
           (np.array([0, 1, 1]), None),
           (np.array([0, 1, 1, 0]), np.array([1, 1, 1])),
           (np.array([0, 1, 1, 0]), np.array([1, 0, 1, 1])),
           (np.array([0, 2, 1, 0]), np.array([1, 1, 1, 1]))]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for ns, nt in ok:
            m, r = LogisticModel(ns, nt, _D()), RefLogit(ns, nt, _D())
            assert np.array_equal(m.n_success, r.n_success)
            assert np.array_equal(m.n_trial, r.n_trial)
            assert m.n_trial.dtype == r.n_trial.dtype == np.float64
        for ns, nt in bad:
            for cls in (LogisticModel, RefLogit):
                with pytest.raises(ValueError):
                    cls(ns, nt, _D())
