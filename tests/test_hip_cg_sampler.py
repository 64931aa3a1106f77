"""GPU parity of the CG sampler: HipCGSampler.sample (-> bbx_cg_sample) vs the
CPU oracle (oracle.cg_sample) on identical inputs, Gaussian draws included.

Tolerance (stated, floating point): the reference's CPU-vs-GPU test accepts
atol=1e-5 on coefficients (tests/gpu_tests/test_gibbs.py:44).  CG stops at
||r|| < 1e-5 sqrt(P) in preconditioned coordinates, so two correct
implementations can differ by about that much times s (rounding differences of
the f64 sums are amplified by the CG recurrence on these ill-conditioned
systems); we require the tighter 1e-6 * max(1, max|coef|) when both take the
same number of iterations, and allow n_iter to differ by one when ||r|| grazes
the threshold (then the reference's own 1e-5)."""
import numpy as np
import pytest

import oracle
from helpers import cg_inputs, mixed_design

pytestmark = pytest.mark.gpu


def _run_both(X, inputs, center=True, intercept=True, maxiter=500,
              storage='csr'):
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix
    hip = HipSparseDesignMatrix(X, center_predictor=center,
                                add_intercept=intercept, storage=storage)
    ora = oracle.OracleSparseDesign(X, center_predictor=center,
                                    add_intercept=intercept)
    n, P = ora.shape
    atol = 10e-6 * np.sqrt(P)            # reg_coef_sampler.py:95
    c_o, i_o = oracle.cg_sample(
        ora, inputs['obs_prec'], inputs['prior_prec_sqrt'], inputs['z'],
        inputs['coef_cg_init'], inputs['coef_scaled_sd'],
        inputs['n_unshrunk'], inputs['randn_n'], inputs['randn_P'], maxiter,
        atol)
    # The HIP wrapper draws eta from the global NumPy stream exactly like the
    # reference (cg_sampler.py:61-62); plant the same values there.
    class _Replay:
        def __init__(self, vecs): self.vecs = list(vecs)
        def __call__(self, size): return self.vecs.pop(0)
    orig = np.random.randn
    np.random.randn = _Replay([inputs['randn_n'], inputs['randn_P']])
    try:
        c_h, i_h = HipCGSampler(inputs['n_unshrunk']).sample(
            hip, inputs['obs_prec'], inputs['prior_prec_sqrt'], inputs['z'],
            coef_cg_init=inputs['coef_cg_init'], precond_by='prior',
            coef_scaled_sd=inputs['coef_scaled_sd'], maxiter=maxiter,
            atol=atol)
    finally:
        np.random.randn = orig
    return c_h, i_h, c_o, i_o


def _assert_close(c_h, i_h, c_o, i_o):
    assert i_h['converged'] == i_o['converged']
    assert i_h['valid_input']
    # the stopping iteration can move by up to 2 when ||r|| hovers around atol
    # (a 1e-15 perturbation of Omega moves the CPU oracle itself by 2, see
    # DESIGN.md "Tolerances"); the coefficients then agree to the reference's
    # own CPU-vs-GPU bound instead of 1e-6
    # (long solves sit on a flat stretch of the residual curve for several
    # iterations: 4 % of the iteration count there)
    assert abs(i_h['n_iter'] - i_o['n_iter']) <= max(2, i_o['n_iter'] // 25)
    scale = np.abs(c_o).max()
    tol = 1e-6 if i_h['n_iter'] == i_o['n_iter'] else 1e-5
    assert np.abs(c_h - c_o).max() <= tol * max(scale, 1.)


@pytest.mark.parametrize("storage", ['csr', 'tiled'])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cg_sample_small_mixed(seed, storage):
    X = mixed_design(300, 40, binary_frac=.5, seed=seed)
    n, P = X.shape[0], X.shape[1] + 1
    out = _run_both(X, cg_inputs(n, P, seed=seed), storage=storage)
    _assert_close(*out)


def test_cg_sample_no_intercept_two_unshrunk():
    X = mixed_design(400, 30, binary_frac=.3, seed=4)
    n, P = X.shape
    inp = cg_inputs(n, P, n_unshrunk=2, seed=4, flat_intercept=False)
    out = _run_both(X, inp, center=False, intercept=False)
    _assert_close(*out)


@pytest.mark.parametrize("storage", ['csr', 'tiled'])
def test_cg_sample_binary_medium(storage):
    from bayesbridge_amd import simulate
    X = simulate.simulate_binary_csr_fast(20000, 2000, .01, seed=5)
    n, P = X.shape[0], X.shape[1] + 1
    out = _run_both(X, cg_inputs(n, P, seed=5), storage=storage)
    _assert_close(*out)


def test_cg_maxiter_exhausted_warns():
    X = mixed_design(300, 40, binary_frac=.5, seed=6)
    n, P = X.shape[0], X.shape[1] + 1
    with pytest.warns(UserWarning):
        c_h, i_h, c_o, i_o = _run_both(X, cg_inputs(n, P, seed=6), maxiter=3)
    assert not i_h['converged'] and not i_o['converged']
    assert i_h['n_iter'] == i_o['n_iter'] == 3
    assert np.abs(c_h - c_o).max() <= 1e-9 * max(1., np.abs(c_o).max())


def test_cg_zero_warm_start_and_device_rng():
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix
    X = mixed_design(2000, 100, binary_frac=.8, seed=7)
    n, P = X.shape[0], X.shape[1] + 1
    inp = cg_inputs(n, P, seed=7)
    inp['coef_cg_init'] = np.zeros(P)
    out = _run_both(X, inp)
    _assert_close(*out)
    # device-side Philox draws: different stream, same distribution.  The
    # mean of many draws approaches Sigma z (checked loosely), and two calls
    # with the same device seed are bitwise identical.
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True)
    sampler = HipCGSampler(1)
    kw = dict(coef_cg_init=inp['coef_cg_init'], precond_by='prior',
              coef_scaled_sd=inp['coef_scaled_sd'], maxiter=500,
              atol=10e-6 * np.sqrt(P))
    a, _ = sampler.sample(hip, inp['obs_prec'], inp['prior_prec_sqrt'],
                          inp['z'], device_rng_seed=123, **kw)
    b, _ = sampler.sample(hip, inp['obs_prec'], inp['prior_prec_sqrt'],
                          inp['z'], device_rng_seed=123, **kw)
    c, _ = sampler.sample(hip, inp['obs_prec'], inp['prior_prec_sqrt'],
                          inp['z'], device_rng_seed=124, **kw)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, c)


@pytest.mark.parametrize("n,p,dtype", [(8192, 200, 'float32'),
                                       (4500, 4300, 'float32'),
                                       (6000, 700, 'float64')])
def test_cg_sample_dense_single_pass_operator(n, p, dtype):
    """Dense designs (f32 storage up to 8192 columns, f64 up to 4096) run
    the CG operator in ONE pass over the matrix (dense_fused_kernel, one and
    two column groups per thread).  The
    matrix is made exactly representable in f32 so that the f64 oracle sees
    the same numbers."""
    from bayesbridge_amd import HipCGSampler, HipDenseDesignMatrix
    rng = np.random.default_rng(n + p)
    X = rng.standard_normal((n, p)).astype(np.float32).astype(np.float64)
    inp = cg_inputs(n, p + 1, seed=3)
    ora = oracle.OracleDenseDesign(X, center_predictor=False,
                                   add_intercept=True)
    atol = 10e-6 * np.sqrt(p + 1)
    c_o, i_o = oracle.cg_sample(
        ora, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
        inp['coef_cg_init'], inp['coef_scaled_sd'], inp['n_unshrunk'],
        inp['randn_n'], inp['randn_P'], 500, atol)
    hip = HipDenseDesignMatrix(X, center_predictor=False, add_intercept=True,
                               storage_dtype=dtype)

    class _Replay:
        def __init__(self, vecs): self.vecs = list(vecs)
        def __call__(self, size): return self.vecs.pop(0)
    orig = np.random.randn
    np.random.randn = _Replay([inp['randn_n'], inp['randn_P']])
    try:
        c_h, i_h = HipCGSampler(inp['n_unshrunk']).sample(
            hip, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
            coef_cg_init=inp['coef_cg_init'], precond_by='prior',
            coef_scaled_sd=inp['coef_scaled_sd'], maxiter=500, atol=atol)
    finally:
        np.random.randn = orig
    _assert_close(c_h, i_h, c_o, i_o)
    # matvec counters: the single pass counts as one dot and one Tdot, and the
    # initial residual of this warm start is one such pass (TD_RESID)
    assert hip.get_dot_count()[0] == hip.get_dot_count()[1] == i_h['n_iter'] + 1


@pytest.mark.parametrize("n,p", [(20000, 2000), (3000, 300)])
def test_update_in_the_tdot_epilogue_equals_the_separate_update(n, p):
    """The tiled layout folds `alpha = rho / p.Ap; x += alpha p; r -= alpha q`
    into the Tdot epilogue, with p.Ap = <p, d p> + <t, Omega t>
    (cg_sampler.hip, apply_operator).  The reference layout cannot deliver
    <t, Omega t> from its dot kernel and runs SciPy's literal `dotprod(p, q)`
    in cg_update_kernel.  Same recurrence on the same matrix, the curvature
    (and the products' summation order) rounded differently in the last bits:
      * stopped after 1 iteration the iterates agree to 1e-12 of their scale
        (the algebra is exact), after 3 to 1e-8 (the flat-prior intercept
        gives the preconditioned system an eigenvalue ~1e5 times the others;
        a last-bit difference in alpha comes back multiplied by it);
      * run to convergence both stop within two iterations of each other and
        agree to the bound this file uses against the oracle."""
    import warnings
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix
    X = mixed_design(n, p, binary_frac=.8, seed=3)
    inp = cg_inputs(n, p + 1, seed=3, lam_log_sd=.3)

    def draw(storage, maxiter):
        hip = HipSparseDesignMatrix(X, center_predictor=True,
                                    add_intercept=True, storage=storage)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')   # short runs stop at maxiter
            coef, info = HipCGSampler(inp['n_unshrunk']).sample(
                hip, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
                coef_cg_init=inp['coef_cg_init'], precond_by='prior',
                coef_scaled_sd=inp['coef_scaled_sd'], maxiter=maxiter,
                atol=10e-6 * np.sqrt(p + 1), seed=10)
        return coef, info, hip.get_dot_count()
    for maxiter, tol in ((1, 1e-12), (3, 1e-8)):
        (a, ia, ca), (b, ib, cb) = draw('csr', maxiter), draw('tiled', maxiter)
        assert ia['n_iter'] == ib['n_iter'] == maxiter
        assert ca == cb
        assert np.abs(a - b).max() <= tol * max(1., np.abs(a).max()), maxiter
    (a, ia, _), (b, ib, _) = draw('csr', 500), draw('tiled', 500)
    assert ia['converged'] and ib['converged']
    assert abs(ia['n_iter'] - ib['n_iter']) <= 2
    tol = 1e-6 if ia['n_iter'] == ib['n_iter'] else 1e-5
    assert np.abs(a - b).max() <= tol * max(1., np.abs(a).max())


def test_cg_sample_wide_design_with_column_groups_in_the_dot():
    """p = 40 000 > one LDS slice: the tiled X~ v runs with several column
    groups per row panel (G > 1, partial slabs + tiled_dot_finalize_kernel),
    a shape whose dot kernel does not deliver <t, Omega t>; the CG loop then
    keeps cg_update_kernel (the fallback of apply_operator).  Against the
    oracle like every other case."""
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    X = simulate.simulate_binary_csr_fast(3000, 40000, .003, seed=5)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    assert hip.tiled_info()['X']['G'] > 1
    del hip
    n, P = X.shape[0], X.shape[1] + 1
    out = _run_both(X, cg_inputs(n, P, seed=2, lam_log_sd=.3), storage='tiled')
    _assert_close(*out)


def test_cg_sample_with_more_row_panels_than_partial_sum_slots(monkeypatch):
    """Past ~1M rows the X layout has more row panels (workgroups) than the 256
    partial-sum slots of bbx_design::part: the X~ v kernel then leaves the sums
    of t and <t, Omega t> to separate kernels and the CG loop keeps
    cg_update_kernel (launch_dot_tiled: `n_panel <= NPART`).  Reached here at
    40 000 rows by forcing 128-row panels (313 workgroups); against the oracle,
    and against the same design with the builder's own panels."""
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    X = simulate.simulate_binary_csr_fast(40000, 1500, .02, seed=6)
    n, P = X.shape[0], X.shape[1] + 1
    inputs = cg_inputs(n, P, seed=3, lam_log_sd=.3)
    plain = _run_both(X, inputs, storage='tiled')
    monkeypatch.setenv('BBX_TILED_PR', '128')
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    assert hip.tiled_info()['X']['grid'] > 256
    del hip
    forced = _run_both(X, inputs, storage='tiled')
    monkeypatch.delenv('BBX_TILED_PR')
    _assert_close(*forced)
    assert abs(forced[1]['n_iter'] - plain[1]['n_iter']) <= 1
    assert np.abs(forced[0] - plain[0]).max() <= 1e-6 * max(
        1., np.abs(plain[0]).max())


@pytest.mark.parametrize("shape", [(6000, 900, .05), (9000, 20000, .004),
                                   (20000, 1000, .02),
                                   (200000, 30000, .002)])
def test_folded_direction_step_is_the_same_solve(shape):
    """bbx_design_set_cg_fold(h, 1): three launches per CG iteration -- the
    stop test, beta and t_k = X~ (s.*r_k) + beta t_{k-1} inside the X~ v kernel
    (csrc/common.hpp DotFold; one and several column blocks) -- against the
    oracle and against the four-launch loop on the same design: same iteration count, same
    draw to rounding; bitwise reproducible; warm and cold start; a solve cut
    off at maxiter reports what the default loop reports."""
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix, simulate
    n, p, f = shape
    X = simulate.simulate_binary_csr_fast(n, p, f, seed=21)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    ora = oracle.OracleSparseDesign(X, center_predictor=True,
                                    add_intercept=True)
    P = p + 1
    atol = 10e-6 * np.sqrt(P)
    if hip.tiled_info()['X']['G'] != 1:
        # several column groups per panel: the X~ v product is two kernels and
        # the request changes nothing (5 launches: direction, dot + its
        # finalize ... the update stays separate)
        before = hip.cg_launches
        hip.set_cg_fold(True)
        assert hip.cg_launches == before == 5
        return
    # default: folded up to 250 000 rows (where it measures faster)
    assert hip.cg_launches == (3 if n <= 250000 else 4)
    hip.set_cg_fold(False)
    assert hip.cg_launches == 4
    for seed, cold in ((3, False), (4, True)):
        inp = cg_inputs(n, P, seed=seed)
        if cold:
            inp['coef_cg_init'] = np.zeros(P)
        c_o, i_o = oracle.cg_sample(
            ora, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
            inp['coef_cg_init'], inp['coef_scaled_sd'], 1, inp['randn_n'],
            inp['randn_P'], 500, atol)

        def draw(maxiter=500):
            class _Replay:
                def __init__(self, vecs): self.vecs = list(vecs)
                def __call__(self, size): return self.vecs.pop(0)
            orig = np.random.randn
            np.random.randn = _Replay([inp['randn_n'], inp['randn_P']])
            try:
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    return HipCGSampler(1).sample(
                        hip, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
                        coef_cg_init=inp['coef_cg_init'], precond_by='prior',
                        coef_scaled_sd=inp['coef_scaled_sd'], maxiter=maxiter,
                        atol=atol)
            finally:
                np.random.randn = orig
        hip.set_cg_fold(False)
        c4, i4 = draw()
        c4_cut, i4_cut = draw(maxiter=3)
        hip.set_cg_fold(True)
        assert hip.cg_launches == 3
        hip.reset_matvec_count()
        c3, i3 = draw()
        counts = hip.get_dot_count()
        c3_again, i3_again = draw()
        c3_cut, i3_cut = draw(maxiter=3)
        hip.set_cg_fold(None)
        _assert_close(c3, i3, c_o, i_o)
        _assert_close(c4, i4, c_o, i_o)
        # (long solves sit on a flat stretch of the residual curve, see
        # _assert_close: each loop is within 4 % of the oracle's count there,
        # so the two are within 8 % of each other -- 83 and 89 iterations
        # around the oracle's 86 on the 20 000 x 1 000 design)
        assert abs(i3['n_iter'] - i4['n_iter']) <= 2 * max(1, i4['n_iter'] // 25)
        # (each is within 1e-6 of the oracle at equal counts: s.*p is formed as
        # s.*r + beta s.*p_old instead of s.*(r + beta p), a rounding-level
        # change that the recurrence carries along)
        scale = max(1., np.abs(c4).max())
        tol = 2e-6 if i3['n_iter'] == i4['n_iter'] else 1e-5
        assert np.abs(c3 - c4).max() <= tol * scale
        assert np.array_equal(c3, c3_again) and i3 == i3_again
        # products that ran: n_iter applications, X~ (s x0) for a warm start,
        # one transposed product for the initial residual
        warm = 0 if cold else 1
        assert counts == (i3['n_iter'] + warm, i3['n_iter'] + 1)
        assert i3_cut['n_iter'] == i4_cut['n_iter'] == 3
        assert not i3_cut['converged'] and not i4_cut['converged']
        assert np.abs(c3_cut - c4_cut).max() <= 1e-9 * max(
            1., np.abs(c4_cut).max())


@pytest.mark.parametrize("fold", [True, False])
def test_long_solves_meet_atol_on_the_recomputed_residual(fold):
    """The folded direction step (default up to 250 000 rows) replaces
    t = X~ (s.*p) by the recurrence t_k = X~ (s.*r_k) + beta t_{k-1}, never
    recomputed from p: what is asserted elsewhere is closeness to the oracle
    and the RECURSIVELY updated residual.  Here the true residual of the
    returned draw, b - A x rebuilt from scratch with separate operator calls
    (cg_sampler.py:66-80,104-109), on ill-conditioned systems -- local scales
    over four decades, 80+ iterations -- must meet the tolerance up to a small
    factor (accumulated rounding of the recurrences), with the fold and
    without it."""
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix, simulate
    n, p = 60000, 6000
    X = simulate.simulate_binary_csr_fast(n, p, .01, seed=31)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    if hip.tiled_info()['X']['G'] != 1:
        pytest.skip("one column group per panel is what the fold needs")
    hip.set_cg_fold(fold)
    assert hip.cg_launches == (3 if fold else 4)
    P = p + 1
    atol = 10e-6 * np.sqrt(P)
    worst = 0.
    for seed in (1, 2, 3):
        inp = cg_inputs(n, P, seed=seed, lam_log_sd=2.3)
        omega, phi, z = inp['obs_prec'], inp['prior_prec_sqrt'], inp['z']
        sd = inp['coef_scaled_sd']
        draw_seed = 50 + seed
        coef, info = HipCGSampler(1).sample(
            hip, omega, phi, z, coef_cg_init=inp['coef_cg_init'],
            precond_by='prior', coef_scaled_sd=sd, maxiter=500, atol=atol,
            seed=draw_seed)
        assert info['converged'] and info['n_iter'] >= 80, info
        np.random.seed(draw_seed)
        eta1, eta2 = np.random.randn(n), np.random.randn(P)
        b = z + hip.Tdot(np.sqrt(omega) * eta1) + phi * eta2
        s = np.empty(P)
        s[0] = 2. * sd[0]                               # cg_sampler.py:128-138
        s[1:] = 1. / phi[1:]
        resid = s * (b - (hip.Tdot(omega * hip.dot(coef)) + phi ** 2 * coef))
        worst = max(worst, float(np.linalg.norm(resid)) / atol)
    # the recurrence's residual passed ||r|| < atol; the recomputed one differs
    # by the rounding accumulated over 80-200 iterations
    assert worst <= 3., (fold, worst)
