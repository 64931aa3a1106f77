"""CPU: the C-ABI library loads and exports every symbol include/bbx.h
declares (no compute without a GPU), and the host-side logic of the product
(prior, summaries, options, synthetic generator) behaves like the reference's."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bbx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bbx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = _declared_symbols()
    assert len(names) >= 40
    from bayesbridge_amd import _lib, hostrng
    lib = _lib.load()
    host = hostrng.load()
    for name in names:
        owner = host if name.startswith("bbx_host_") else lib
        assert hasattr(owner, name), name
    assert lib.bbx_version() >= 100
    # every symbol the ctypes layer binds is declared in the header
    assert set(_lib.EXPORTED_SYMBOLS) <= set(names)


def test_no_gpu_means_loud_failure():
    from bayesbridge_amd import _lib
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    import scipy.sparse as sparse
    from bayesbridge_amd import BbxError, HipSparseDesignMatrix
    with pytest.raises(BbxError):
        HipSparseDesignMatrix(sparse.random(20, 5, density=.5, format='csr'))
    h = ctypes.c_void_p()
    st = _lib.load().bbx_design_create_csr(
        10, 2, 0, None, None, None, None, 1, 0, 0, ctypes.byref(h))
    assert st < 0 and _lib.last_error()


def test_prior_matches_reference_formulas():
    from bayesbridge_amd import RegressionCoefPrior
    p = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    assert p.param['gscale_neg_power'] == {'shape': 0., 'rate': 0.}
    unit = math.gamma(2 / .5) / math.gamma(1 / .5)      # prior.py:163-167
    assert p.compute_power_exp_ave_magnitude(.5) == unit
    g, l = p.adjust_scale(1., np.ones(3), to='raw')
    assert g == 1. / unit and np.all(l == unit)
    q = RegressionCoefPrior(
        bridge_exponent=.25,
        global_scale_prior_hyper_param={'log10_mean': -4., 'log10_sd': 1.})
    shape, rate = (q.param['gscale_neg_power'][k] for k in ('shape', 'rate'))
    from scipy.special import polygamma
    # the solve of prior.py:169-199: sd and mean of log(tau^-alpha) match
    assert abs(math.sqrt(float(polygamma(1, shape))) / .25
               - math.log(10.)) < 1e-8
    assert shape > 0 and rate > 0
    with pytest.raises(ValueError):
        RegressionCoefPrior(bridge_exponent=3.)
    assert q.clone(bridge_exponent=.5).bridge_exp == .5


def test_summarizer_equals_oracle_summarizer():
    from bayesbridge_amd.reg_coef_sampler import \
        RegressionCoeffficientPosteriorSummarizer
    from oracle.summarizer import CoefSummarizer
    rng = np.random.default_rng(0)
    a = RegressionCoeffficientPosteriorSummarizer(7, 1, 2.)
    b = CoefSummarizer(7, 1, 2.)
    for _ in range(5):
        coef, g, l = rng.standard_normal(7), rng.random() + .1, \
            rng.random(6) + .1
        assert np.allclose(a.extrapolate_coef_condmean(g, l),
                           b.extrapolate_coef_condmean(g, l))
        assert np.allclose(a.estimate_coef_precond_scale_sd(),
                           b.estimate_post_sd())
        a.update(coef, g, l)
        b.update(coef, g, l)


def test_sampler_options_reject_non_cg():
    from bayesbridge_amd import SamplerOptions

    class _D:
        shape = (10, 3)
    for bad in ('cholesky', 'hmc'):
        with pytest.raises(ValueError):      # gibbs_util.py:49-50 analogue
            SamplerOptions.pick_default_and_create(bad, None, 'logit', _D())
    with pytest.raises(ValueError):
        SamplerOptions.pick_default_and_create('nuts', None, 'logit', _D())
    opt = SamplerOptions.pick_default_and_create(None, None, 'logit', _D())
    assert opt.coef_sampler_type == 'cg' and opt.rng == 'device'
    with pytest.raises(ValueError):
        SamplerOptions(rng='cuda')


def test_fast_generator_distribution():
    from bayesbridge_amd import simulate
    X = simulate.simulate_binary_csr_fast(20000, 500, .02, seed=4)
    assert np.all(X.data == 1.)
    col = np.bincount(X.indices, minlength=500)
    assert col.max() <= .5 * 20000 + 1          # max_freq_per_col = .5
    assert .5 < col.mean() / (.02 * 20000) < 1.6
    assert X.has_sorted_indices
    # exactly distinct rows per column
    assert (X.T.tocsr().multiply(X.T.tocsr()) != X.T.tocsr()).nnz == 0


def test_device_forms_of_the_polya_gamma_pieces_equal_the_reference_forms():
    """The device chain's Polya-Gamma kernel (csrc/pg_queue.hpp) forms the
    mixture weight of the exponential piece and the terms of the alternating
    series directly -- products and one exp per term -- where the reference
    sums logarithms (random/polya_gamma/polya_gamma.pyx:115-137).  Both forms
    live in csrc/samplers.hpp and compile for the host: here they are compared
    where it can be done exactly, without a GPU.  The weight agrees to a few
    ulps across z = |psi| / 2 in [0, 20] (beyond, the kernel calls the log form
    itself); the series test takes the same decision for every (proposal,
    uniform) pair of a grid -- except, possibly, uniforms within rounding of a
    partial sum, which the grid's 2e5 pairs never hit."""
    from ctypes import c_void_p
    from bayesbridge_amd import hostrng
    lib = hostrng.load()
    z = np.concatenate([np.linspace(0., 20., 4001), np.array([20.5, 30., 60.]),
                        np.random.default_rng(0).random(2000) * 2.])
    a, b = np.empty_like(z), np.empty_like(z)
    assert lib.bbx_host_pg_right_mass(
        len(z), c_void_p(z.ctypes.data), c_void_p(a.ctypes.data),
        c_void_p(b.ctypes.data)) == 0
    assert np.all((a >= 0) & (a < 1)) and np.all(np.isfinite(b))
    assert np.all(a[z <= 30.] > 0)      # (at z = 60 the weight underflows: both 0)
    # relative error (the weight falls to 1e-50 at z = 20)
    ok = a > 0
    assert np.max(np.abs(a[ok] - b[ok]) / a[ok]) < 1e-12
    assert np.array_equal(a[z > 20.], b[z > 20.])      # the log form itself
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.random(100000) * .64, .64 + rng.exponential(.5, 100000),
                        np.array([2. / np.pi, 1e-3, 1e-2, 25.])])
    u = rng.random(len(x))
    u[:2000] = 1. - 1e-4 * rng.random(2000)          # near the first partial sum
    s, d = np.empty(len(x), dtype=np.int32), np.empty(len(x), dtype=np.int32)
    assert lib.bbx_host_pg_series_accept(
        len(x), c_void_p(x.ctypes.data), c_void_p(u.ctypes.data),
        c_void_p(s.ctypes.data), c_void_p(d.ctypes.data)) == 0
    assert np.array_equal(s, d)
    assert .9 < s.mean() <= 1.      # the envelope is tight, and rejections exist
    assert s.min() == 0
