"""CPU: the oracle restatement against the committed golden vectors (captured
from the imported reference by tests/golden/make_golden.py) and against the
reference's own regression fixtures.  No GPU, no /root/reference needed."""
import os

import numpy as np
import pytest
import scipy.sparse as sparse

import oracle
from oracle.gibbs import OracleGibbs


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_operator_sparse_matches_reference_outputs(golden_dir):
    g = _load(golden_dir, 'operator_sparse_100x10.npz')
    d = oracle.OracleSparseDesign(sparse.csr_matrix(g['X']),
                                  center_predictor=True, add_intercept=True)
    assert np.abs(d.dot(g['v']) - g['dot']).max() <= 1e-13
    assert np.abs(d.Tdot(g['w']) - g['Tdot']).max() <= 1e-12
    # explicit matrix, as tests/test_design_matrix.py:12-24 checks it
    A = d.toarray()
    assert np.allclose(A.dot(g['v']), g['dot'], atol=10e-6, rtol=10e-6)
    assert np.allclose(A.T.dot(g['w']), g['Tdot'], atol=10e-6, rtol=10e-6)


def test_operator_dense_matches_reference_outputs(golden_dir):
    g = _load(golden_dir, 'operator_dense_100x10.npz')
    d = oracle.OracleDenseDesign(g['X'], center_predictor=True,
                                 add_intercept=True)
    assert np.abs(d.dot(g['v']) - g['dot']).max() <= 1e-12
    assert np.abs(d.Tdot(g['w']) - g['Tdot']).max() <= 1e-11


@pytest.mark.parametrize("name,fmt", [('chain_linear_dense_cg.npz', 'dense'),
                                      ('chain_logit_sparse_cg.npz', 'sparse')])
@pytest.mark.parametrize("use_scipy", [False, True])
def test_cg_sample_replays_reference_iterations(golden_dir, name, fmt,
                                                use_scipy):
    """Every recorded call of the reference's ConjugateGradientSampler.sample
    (inputs incl. its two Gaussian vectors) is replayed through the oracle."""
    g = _load(golden_dir, name)
    X = sparse.csr_matrix(g['X']) if fmt == 'sparse' else g['X']
    d = oracle.make_design(X)
    for it in range(g['cg_coef'].shape[0]):
        coef, info = oracle.cg_sample(
            d, g['cg_obs_prec'][it], g['cg_prior_prec_sqrt'][it], g['cg_z'][it],
            g['cg_coef_cg_init'][it], g['cg_coef_scaled_sd'][it],
            int(g['cg_n_unshrunk'][it]), g['cg_randn_n'][it],
            g['cg_randn_P'][it], int(g['cg_maxiter'][it]),
            float(g['cg_atol'][it]), use_scipy=use_scipy)
        assert info['n_iter'] == int(g['cg_n_iter'][it])
        assert np.abs(coef - g['cg_coef'][it]).max() <= 1e-10


@pytest.mark.parametrize("model,fmt", [('linear', 'dense'),
                                       ('logit', 'sparse')])
def test_chain_reproduces_reference_golden_vectors(golden_dir, model, fmt):
    """tests/regression_tests/test_gibb.py:11-58,107-109 through the oracle:
    the last of 10 samples must match the reference's saved output within its
    own tolerance (rtol=1e-3, atol=1e-5)."""
    g = _load(golden_dir, 'chain_%s_%s_cg.npz' % (model, fmt))
    saved = _load(golden_dir, 'reference_%s_cg_last_sample.npy' % model)
    X = sparse.csr_matrix(g['X']) if fmt == 'sparse' else g['X']
    outcome = g['y'] if model == 'linear' else (g['n_success'], g['n_trial'])
    chain = OracleGibbs(outcome, X, model, bridge_exponent=.25,
                        sd_for_intercept=2., regularizing_slab_size=1.)
    res = chain.gibbs(10, seed=0, init={'global_scale': .1,
                                        'local_scale': np.ones(50)})
    assert np.allclose(res['coef'][:, -1], saved, rtol=.001, atol=10e-6)
    # and the whole recorded chain of the imported reference, much tighter
    # (not bitwise: the oracle's log Phi uses libm erfc where the reference
    # vendors Cephes, a last-ulp difference in Omega that CG amplifies)
    assert np.array_equal(res['n_cg_iter'], g['n_cg_iter'])
    assert np.abs(res['coef'] - g['coef_samples']).max() <= 1e-6
    assert np.allclose(res['global_scale'], g['global_scale_samples'],
                       rtol=1e-6)
    assert np.allclose(res['logp'], g['logp_samples'], rtol=1e-6)


def test_chain_mixed_logit_initcoef(golden_dir):
    """tests/gpu_tests/test_gibbs.py:34-44 inputs on the CPU path."""
    g = _load(golden_dir, 'chain_logit_mixed_initcoef.npz')
    X = sparse.csr_matrix((g['X_data'], g['X_indices'], g['X_indptr']),
                          shape=tuple(g['X_shape']))
    chain = OracleGibbs((g['n_success'], g['n_trial']), X, 'logit')
    res = chain.gibbs(10, seed=1, init={'coef': np.ones(X.shape[1] + 1)})
    # harder systems (70-90 CG iterations): a last-ulp difference in Omega can
    # move the stopping iteration by a few; the coefficients stay within the
    # reference's own CPU-vs-GPU tolerance (tests/gpu_tests/test_gibbs.py:44)
    assert np.abs(res['n_cg_iter'] - g['n_cg_iter']).max() <= 5
    assert np.allclose(res['coef'], g['coef_samples'], atol=1e-5)


def test_chain_config2_scaled_binary_logit(golden_dir):
    """BASELINE config 2 scaled down (20000 x 1000 binary CSR, logit, demo
    prior): the oracle on the regenerated design reproduces the reference's
    first 10 samples."""
    from helpers import config2_small_problem
    g, X, outcome = config2_small_problem(golden_dir)
    chain = OracleGibbs(outcome, X, 'logit', bridge_exponent=.5,
                        regularizing_slab_size=2.)
    res = chain.gibbs(10, seed=111, init={'global_scale': .01})
    assert np.abs(res['n_cg_iter'] - g['n_cg_iter'][:10]).max() <= 2
    assert np.allclose(res['coef'], g['coef_first10'], atol=1e-5)
    assert np.allclose(res['global_scale'], g['global_scale_first10'],
                       rtol=1e-6)
    assert np.allclose(res['logp'], g['logp_first10'], rtol=1e-6)


def test_scipy_style_cg_equals_scipy():
    import scipy.sparse.linalg as spla
    rng = np.random.default_rng(3)
    M = rng.standard_normal((60, 40))
    A = M.T @ M + np.eye(40)
    b = rng.standard_normal(40)
    x0 = rng.standard_normal(40)
    for maxiter in (3, 200):
        n_cb = [0]
        xs, info_s = spla.cg(A, b, x0=x0, rtol=1e-8, atol=0., maxiter=maxiter,
                             callback=lambda x: n_cb.__setitem__(0, n_cb[0] + 1))
        n_mine = [0]
        xm, info_m = oracle.scipy_style_cg(
            lambda v: A @ v, b, x0, 1e-8, 0., maxiter,
            lambda x: n_mine.__setitem__(0, n_mine[0] + 1))
        assert info_s == info_m and n_cb[0] == n_mine[0]
        assert np.abs(xs - xm).max() <= 1e-12 * max(1., np.abs(xs).max())


def test_c_csr_products_match_scipy():
    from oracle import rng as orng
    X = sparse.random(300, 70, density=.1, random_state=1, format='csr')
    v = np.random.default_rng(0).standard_normal(70)
    w = np.random.default_rng(1).standard_normal(300)
    assert np.abs(orng.csr_matvec(X, v) - X @ v).max() <= 1e-12
    assert np.abs(orng.csr_rmatvec(X, w) - X.T @ w).max() <= 1e-12


def config4_small_problem(golden_dir):
    """The scaled BASELINE config 4 problem of
    chain_linear_dense_4000x800_f32repr.npz (design regenerated, checked
    against the fixture's heads and sums)."""
    from bayesbridge_amd import simulate
    g = _load(golden_dir, 'chain_linear_dense_4000x800_f32repr.npz')
    n, p = (int(v) for v in g['shape'])
    np.random.seed(111)
    X = np.random.randn(n, p).astype(np.float32).astype(np.float64)
    assert np.array_equal(X[:4, :4], g['X_head']) and X.sum() == g['X_sum']
    y = simulate.simulate_outcome(X, simulate.demo_beta(p), 'linear', seed=1)
    assert np.allclose(y[:8], g['y_head'], rtol=1e-14)
    assert abs(y.sum() - g['y_sum']) <= 1e-9 * abs(g['y_sum'])
    return g, X, y


def test_config4_scaled_chain_through_the_oracle(golden_dir):
    """Linear model, dense 4000 x 800 (config 4 scaled): the oracle chain with
    the L-BFGS mode search reproduces the reference's 10 samples."""
    g, X, y = config4_small_problem(golden_dir)
    chain = OracleGibbs(y, X, 'linear', bridge_exponent=.5,
                        regularizing_slab_size=2.)
    res = chain.gibbs(10, seed=111, init={'global_scale': .01})
    assert np.abs(res['n_cg_iter'] - g['n_cg_iter']).max() <= 2
    assert np.allclose(res['coef'], g['coef_samples'], atol=1e-5)
    assert np.allclose(res['global_scale'], g['global_scale_samples'],
                       rtol=1e-5)
    assert np.allclose(res['logp'], g['logp_samples'], rtol=1e-6)


def test_chain_config2_full_size_first10(golden_dir):
    """BASELINE config 2 at FULL size (100 000 x 10 000 binary CSR, nnz
    10 221 685, logit, demo prior; tests/golden/make_config2_full.py): the
    design replayed by bayesbridge_amd.simulate carries the reference's
    checksums, and the oracle chain reproduces the reference's first 10 samples
    (256 stored coefficients, sum |coef| over all of them, tau, log posterior,
    n_cg_iter) -- the same code path the GPU test compares the HIP chain with."""
    from helpers import config2_small_problem
    g, X, outcome = config2_small_problem(
        golden_dir, 'chain_logit_binary_100000x10000_first10.npz')
    chain = OracleGibbs(outcome, X, 'logit', bridge_exponent=.5,
                        regularizing_slab_size=2.)
    out = chain.gibbs(10, seed=111, init={'global_scale': .01})
    assert np.allclose(out['coef'][g['picked']], g['coef_first10'], atol=1e-6)
    assert np.allclose(np.abs(out['coef']).sum(axis=0),
                       g['coef_abs_sum_first10'], rtol=1e-7)
    assert np.allclose(out['global_scale'], g['global_scale_first10'],
                       rtol=1e-6)
    assert np.allclose(out['logp'], g['logp_first10'], rtol=1e-8)
    assert np.array_equal(out['n_cg_iter'], g['n_cg_iter'])
