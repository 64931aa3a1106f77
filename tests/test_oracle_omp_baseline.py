"""CPU: the OpenMP multi-core baseline (oracle/csrc/oracle_cg_omp.cpp, the
`cpu_baseline.port_omp` leg of bench.py) against the NumPy/SciPy oracle."""
import numpy as np
import pytest

import oracle
from helpers import cg_inputs, mixed_design
from oracle.omp_baseline import OmpSparseDesign


@pytest.mark.parametrize("binary_frac,threads", [(1., 1), (1., 4), (.5, 3)])
def test_omp_products_and_cg_draw_equal_oracle(binary_frac, threads):
    X = mixed_design(4000, 300, binary_frac=binary_frac, freq=.05, seed=2)
    omp = OmpSparseDesign(X, n_threads=threads)
    ora = oracle.OracleSparseDesign(X, center_predictor=True,
                                    add_intercept=True)
    n, P = ora.shape
    assert omp.shape == (n, P)
    rng = np.random.default_rng(1)
    v, w = rng.standard_normal(P), rng.standard_normal(n)
    assert np.abs(omp.dot(v) - ora.dot(v)).max() <= 1e-11
    assert np.abs(omp.Tdot(w) - ora.Tdot(w)).max() <= 1e-10
    kw = cg_inputs(n, P, seed=4)
    atol = 10e-6 * np.sqrt(P)
    for x0 in (kw['coef_cg_init'], np.zeros(P)):
        c_o, i_o = oracle.cg_sample(
            ora, kw['obs_prec'], kw['prior_prec_sqrt'], kw['z'], x0,
            kw['coef_scaled_sd'], 1, kw['randn_n'], kw['randn_P'], 500, atol)
        c_m, i_m = omp.cg_sample(
            kw['obs_prec'], kw['prior_prec_sqrt'], kw['z'], x0,
            kw['coef_scaled_sd'], 1, kw['randn_n'], kw['randn_P'], 500, atol)
        # +-2, or 5 % on long solves (flat stretch of the residual curve)
        slack = max(2, int(np.ceil(.05 * i_o['n_iter'])))
        assert i_m['converged'] and abs(i_m['n_iter'] - i_o['n_iter']) <= slack
        # both stop at ||r|| < atol; when the stopping iteration moves by one
        # the two solutions differ at the solver's tolerance (the reference's
        # CPU-vs-GPU bound, tests/gpu_tests/test_gibbs.py:44)
        tol = 1e-6 if i_m['n_iter'] == i_o['n_iter'] else 1e-5
        assert np.abs(c_m - c_o).max() <= tol * max(1., np.abs(c_o).max())


def test_oracle_chain_runs_on_the_omp_design():
    from bayesbridge_amd import simulate
    from oracle.gibbs import OracleGibbs
    X = simulate.simulate_design_csr(2000, 100, binary_frac=1.,
                                     binary_pred_freq=.05, seed=5)
    y = simulate.simulate_outcome(X, simulate.demo_beta(100), 'logit', seed=1)
    init = {'global_scale': .05, 'coef': np.zeros(101)}
    a = OracleGibbs(y, X, 'logit', regularizing_slab_size=2.).gibbs(
        4, seed=3, init=dict(init))
    b = OracleGibbs(y, X, 'logit', regularizing_slab_size=2.,
                    omp_threads=2).gibbs(4, seed=3, init=dict(init))
    # same streams, same algorithm; products differ in summation order only
    assert np.abs(a['coef'] - b['coef']).max() <= 1e-5
    assert np.abs(a['n_cg_iter'] - b['n_cg_iter']).max() <= 2
