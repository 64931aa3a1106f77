"""BASELINE config 4 (linear model, dense X 200 000 x 8 000 stored in f32) at
FULL size, plus its scaled-down fixture from the reference.

Full size (6.4 GB of matrix, byte offsets beyond 2^32; the CPU oracle would
need the 12.8 GB f64 matrix and minutes per iteration) is covered by
size-independent properties of dense_matrix.py:37-52 / cg_sampler.py:96-113:
adjointness, linearity, agreement of the single-pass operator kernel with the
two separate products, agreement with torch's f64 matmul on the SAME stored
values, the defining equation of a CG draw, and a reproducible linear-model
device chain.  The scaled problem (4000 x 800, f32-representable entries)
replays the reference's samples through the f32-stored operator.
"""
import os
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, P_MAIN = 200000, 8000


@pytest.fixture(scope="module")
def full_dense():
    import torch
    from bayesbridge_amd import HipDenseDesignMatrix
    gen = torch.Generator(device='cuda')
    gen.manual_seed(111)
    X = torch.randn((N, P_MAIN), generator=gen, device='cuda',
                    dtype=torch.float32)
    offset = X.double().mean(dim=0)
    torch.cuda.synchronize()
    hip = HipDenseDesignMatrix.from_device_array(
        N, P_MAIN, X.data_ptr(), offset.data_ptr(), add_intercept=True,
        device=0, in_dtype='float32', storage_dtype='float32')
    yield hip, X, offset
    del hip


def _stored_rows(X, offset, rows):
    """What the operator holds for these rows: [1 | fl32(x - mean)] in f64."""
    import torch
    centred = (X[rows].double() - offset).float().double()
    ones = torch.ones((centred.shape[0], 1), dtype=torch.float64,
                      device=centred.device)
    return torch.cat([ones, centred], dim=1)


def test_config4_shape_bytes_and_counters(full_dense):
    hip, X, _ = full_dense
    assert hip.shape == (N, P_MAIN + 1) and not hip.is_sparse
    assert hip.storage_format == 'dense'
    dot_b, tdot_b = hip.matvec_bytes
    # 6.401e9 B per GEMV at f32 storage (+ vectors)
    assert 6.40e9 < dot_b < 6.42e9 and 6.40e9 < tdot_b < 6.42e9
    assert 6.40e9 < hip.fused_operator_bytes < 6.43e9
    assert hip.storage_bytes > 2 ** 32          # byte offsets beyond 32 bits


def test_config4_adjoint_linear_and_torch_f64_products(full_dense):
    import torch
    hip, X, offset = full_dense
    n, P = hip.shape
    rng = np.random.default_rng(5)
    v1, v2 = rng.standard_normal(P), rng.standard_normal(P)
    w = rng.standard_normal(n)
    t1 = hip.dot(v1)
    g = hip.Tdot(w)
    lhs, rhs = np.dot(t1, w), np.dot(v1, g)
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.)
    lin = hip.dot(2.5 * v1 + v2) - (2.5 * t1 + hip.dot(v2))
    assert np.abs(lin).max() <= 1e-10 * max(1., np.abs(t1).max())
    # X~ v on a row sample (first, last, random rows) against torch f64
    rows = np.unique(np.concatenate((
        np.arange(64), np.arange(n - 64, n), rng.integers(0, n, 4000))))
    A = _stored_rows(X, offset, torch.from_numpy(rows).cuda())
    ref = (A @ torch.from_numpy(v1).cuda()).cpu().numpy()
    assert np.abs(t1[rows] - ref).max() <= 1e-11 * np.abs(ref).max()
    # X~^T w against torch f64, chunked over all rows
    wd = torch.from_numpy(w).cuda()
    gt = torch.zeros(P, dtype=torch.float64, device='cuda')
    for lo in range(0, n, 20000):
        idx = torch.arange(lo, min(lo + 20000, n), device='cuda')
        gt += _stored_rows(X, offset, idx).T @ wd[idx]
    gt = gt.cpu().numpy()
    assert np.abs(g - gt).max() <= 1e-10 * np.abs(gt).max()


def test_config4_single_pass_operator_equals_two_products(full_dense):
    """dense_fused_kernel (one pass over the 6.4 GB) against Tdot(w * dot(v))
    issued as two separate library calls."""
    hip, *_ = full_dense
    n, P = hip.shape
    rng = np.random.default_rng(9)
    v = rng.standard_normal(P)
    omega = rng.gamma(2., .5, n)
    hip.reset_matvec_count()
    fused = hip.gram_matvec(omega, v)
    assert hip.get_dot_count() == (1, 1)
    two = hip.Tdot(omega * hip.dot(v))
    assert np.abs(fused - two).max() <= 1e-10 * np.abs(two).max()
    # linear in v, and symmetric: <u, G v> = <v, G u>
    u = rng.standard_normal(P)
    gu = hip.gram_matvec(omega, u)
    a, b = np.dot(u, fused), np.dot(v, gu)
    assert abs(a - b) <= 1e-9 * max(abs(a), abs(b))
    both = hip.gram_matvec(omega, 2. * v - u)
    assert np.abs(both - (2. * fused - gu)).max() <= 1e-10 * np.abs(fused).max()


def test_config4_cg_draw_solves_the_perturbed_system(full_dense):
    from bayesbridge_amd import HipCGSampler
    hip, *_ = full_dense
    n, P = hip.shape
    rng = np.random.default_rng(6)
    omega = np.full(n, .9)                          # linear model: obs_prec 1_n
    phi = 1. / (.05 * rng.gamma(1., 1., P) + .01)
    phi[0] = 0.
    z = hip.Tdot(omega * rng.standard_normal(n))
    sd = np.ones(P)
    atol = 1e-5 * np.sqrt(P)
    sampler = HipCGSampler(n_coef_wo_shrinkage=1)
    coef, info = sampler.sample(
        hip, omega, phi, z, coef_cg_init=np.zeros(P), coef_scaled_sd=sd,
        maxiter=500, atol=atol, seed=77)
    assert info['converged'] and 3 <= info['n_iter'] < 500
    np.random.seed(77)
    eta1, eta2 = np.random.randn(n), np.random.randn(P)
    b = z + hip.Tdot(np.sqrt(omega) * eta1) + phi * eta2
    s = np.empty(P)
    s[0] = 2. * sd[0]
    s[1:] = 1. / phi[1:]
    resid = s * (b - (hip.Tdot(omega * hip.dot(coef)) + phi ** 2 * coef))
    assert np.linalg.norm(resid) <= 1.05 * atol
    coef2, info2 = sampler.sample(
        hip, omega, phi, z, coef_cg_init=np.zeros(P), coef_scaled_sd=sd,
        maxiter=500, atol=atol, seed=77)
    assert info2['n_iter'] == info['n_iter'] and np.array_equal(coef, coef2)


def test_config4_linear_device_chain_runs_and_is_reproducible(full_dense):
    import torch
    from bayesbridge_amd import HipGibbsChain
    hip, X, _ = full_dense
    n, P = hip.shape
    beta = torch.zeros(15, dtype=torch.float64, device='cuda')
    beta[:5], beta[5:10], beta[10:15] = 1.5, 1., .5
    gen = torch.Generator(device='cuda')
    gen.manual_seed(1)
    y = (X[:, :15].double() @ beta + torch.randn(
        n, generator=gen, device='cuda', dtype=torch.float64)).cpu().numpy()
    unit = 6.                                   # Gamma(4)/Gamma(2), alpha = .5

    def run(seed):
        chain = HipGibbsChain(hip, 'linear', y, bridge_exponent=.5,
                              slab_size=2., seed=seed)
        coef0 = np.zeros(P)
        coef0[0] = y.mean()
        chain.set_state(coef0, None, np.ones(P - 1) * unit, .01 / unit)
        chain.init_obs_prec()
        kept, n_unconv = chain.run(5, save=('coef', 'obs_prec'))
        ll, lp = chain.logp()
        chain.close()
        return kept, n_unconv, lp
    a, unconv, lp = run(3)
    b, _, lp_b = run(3)
    c, _, _ = run(4)
    assert unconv == 0
    assert np.array_equal(a['coef'], b['coef']) and lp == lp_b   # bitwise
    assert not np.array_equal(a['coef'], c['coef'])
    assert np.all(np.isfinite(a['coef'])) and np.all(np.isfinite(a['logp']))
    assert np.all(a['n_cg_iter'] > 0) and np.all(a['n_cg_iter'] < 500)
    # noise sd is 1: after the first draw (made from coef = 0, where the
    # residual still holds the whole signal) the precision sits at ~1
    assert np.all(a['obs_prec'] > 0.)
    assert np.all(np.abs(a['obs_prec'].ravel()[1:] - 1.) < .2)
    # the five large true effects are found after five iterations
    assert np.all(np.abs(a['coef'][-1][1:6] - 1.5) < .1)


def test_config4_scaled_reference_chain_through_f32_storage(golden_dir):
    """The reference's 10 samples of the scaled config-4 problem (fixture made
    by importing the reference) through dense_storage_dtype='float32' on the
    reference's random streams.  X is f32-representable; the stored CENTRED
    entries are rounded to f32 (6e-8 relative), far inside the reference's
    CPU-vs-GPU bound atol=1e-5 (tests/gpu_tests/test_gibbs.py:44)."""
    from bayesbridge_amd import BayesBridge, RegressionCoefPrior, \
        RegressionModel
    from test_oracle_golden import config4_small_problem
    g, X, y = config4_small_problem(golden_dir)
    prior = RegressionCoefPrior(bridge_exponent=.5, regularizing_slab_size=2.)
    for dtype, tol in (('float32', 1e-5), ('float64', 1e-5)):
        model = RegressionModel(y, X, 'linear', dense_storage_dtype=dtype)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s, info = BayesBridge(model, prior).gibbs(
                10, 0, init={'global_scale': .01}, coef_sampler_type='cg',
                seed=111, params_to_save='all', options={'rng': 'reference'})
        n_cg = info['_reg_coef_sampling_info']['n_cg_iter']
        assert np.abs(n_cg - g['n_cg_iter']).max() <= \
            max(2, int(.05 * g['n_cg_iter'].max()))
        assert np.allclose(s['coef'], g['coef_samples'], atol=tol), dtype
        assert np.allclose(s['global_scale'], g['global_scale_samples'],
                           rtol=1e-5)
        assert np.allclose(s['obs_prec'], g['obs_prec_samples'], rtol=1e-6)
        assert np.allclose(s['logp'], g['logp_samples'], rtol=1e-6)


@pytest.mark.parametrize("shape", [(4097, 801), (20000, 4000), (513, 16)])
def test_matrix_core_gemv_variant_gives_the_same_product(shape):
    """The opt-in MFMA GEMV (dense_dot_mfma_kernel, BBX_DENSE_MFMA=1; the A/B
    of LABNOTES.md 3.3) against torch f64 on the stored values -- the script
    asserts <= 1e-11 relative -- for shapes with ragged row blocks / column
    chunks.  The environment switch is read once per process, hence the
    subprocess."""
    import subprocess
    import sys
    from conftest import ROOT
    for flag in ("1", "0"):
        env = dict(os.environ, BBX_DENSE_MFMA=flag)
        res = subprocess.run(
            [sys.executable, os.path.join(ROOT, "scripts", "ab_dense_mfma.py"),
             str(shape[0]), str(shape[1]), "3"],
            env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        assert "BBX_DENSE_MFMA=%s" % flag in res.stdout


@pytest.mark.parametrize("shape", [(20001, 4001, 'float32'),
                                   (16500, 5000, 'float32'),
                                   (65537, 801, 'float32'),
                                   (20001, 4001, 'float64'),
                                   (16500, 5000, 'float64'),
                                   (33001, 8001, 'float64'),
                                   (65537, 801, 'float64')])
def test_lds_ring_variant_of_the_single_pass_operator_is_bit_identical(shape):
    """dense_fused_ring_kernel (LDS-DMA ring, the default from 64 rows per
    workgroup on) against dense_fused_kernel (register prefetch,
    BBX_DENSE_FUSED_RING=0): same arithmetic in the same order, so the script's
    SHA-256 of the product must agree; it also checks the product against the
    two separate passes (<= 1e-10).  Ragged row ranges, one and two column
    groups per thread; f64 storage: dense_fused_f64_ring_kernel (one row of up
    to 8192 doubles, or two of up to 4096, per ring stage) against
    dense_fused_f64_kernel.  The switch is read once per process:
    subprocesses."""
    import re
    import subprocess
    import sys
    from conftest import ROOT
    digest = {}
    for flag in ("0", "22"):
        env = dict(os.environ, BBX_DENSE_FUSED_RING=flag)
        res = subprocess.run(
            [sys.executable, os.path.join(ROOT, "scripts", "ab_dense_fused.py"),
             str(shape[0]), str(shape[1]), "3", shape[2]],
            env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
        m = re.search(r"BBX_DENSE_FUSED_RING=%s .* sha256 ([0-9a-f]+)" % flag,
                      res.stdout)
        assert m, res.stdout[-2000:]
        digest[flag] = m.group(1)
    assert digest["0"] == digest["22"]


@pytest.mark.parametrize("K", [16, 32])
def test_config4_batched_products_equal_the_single_chain_operator(full_dense, K):
    """The K-column products of dense_batch.hip at full size (X^T W from X, X V
    from the transposed copy; 32 chains: two B operands per A operand) against
    the single-chain kernels of the same handle, column by column, and the
    adjoint identity between the two batched products themselves."""
    from bayesbridge_amd import HipChainBatch, HipGibbsChain
    hip, *_ = full_dense
    n, P = hip.shape
    rng = np.random.default_rng(21)
    y = rng.standard_normal(n)
    batch = HipChainBatch([HipGibbsChain(hip, 'linear', y, sd_unshrunk=[np.inf],
                                         bridge_exponent=.5, slab_size=2.,
                                         seed=s) for s in range(K)], allow_slow=True)
    V, W = rng.standard_normal((K, P)), rng.standard_normal((K, n))
    T, G = batch.dot(V), batch.Tdot(W)
    for c in (0, 7, K - 1):
        t, g = hip.dot(V[c]), hip.Tdot(W[c])
        assert np.abs(T[c] - t).max() <= 1e-11 * np.abs(t).max()
        assert np.abs(G[c] - g).max() <= 1e-10 * np.abs(g).max()
    lhs = np.einsum('cn,cn->c', T, W)       # <X v_c, w_c> = <v_c, X^T w_c>
    rhs = np.einsum('cp,cp->c', V, G)
    assert np.all(np.abs(lhs - rhs) <= 1e-9 * np.maximum(np.abs(lhs), 1.))
    # the transposed copy is accounted for
    assert hip.storage_bytes > 2 * 6.4e9
    # three lock-step Gibbs iterations: every chain converges, chains differ
    samples, n_unconverged = batch.run(3, save_coef=False)
    assert n_unconverged == 0
    assert np.all(samples['n_cg_iter'] > 0)
    assert len(np.unique(samples['logp'][:, -1])) == K


@pytest.mark.parametrize("shape", [(9000, 3000), (12345, 5001), (4100, 8100)])
def test_single_pass_operator_with_f64_storage(shape):
    """The single-pass operator with f64 storage (dense_fused_f64_ring_kernel /
    dense_fused_f64_kernel: two or four pairs of doubles per thread, up to 8192
    stored columns) against the two separate products and NumPy."""
    from bayesbridge_amd import HipDenseDesignMatrix
    n, p = shape
    rng = np.random.default_rng(31)
    X = rng.standard_normal((n, p))
    hip = HipDenseDesignMatrix(X, center_predictor=True, add_intercept=True,
                               storage_dtype='float64')
    assert hip.fused_operator_bytes > 0
    v, omega = rng.standard_normal(p + 1), rng.gamma(2., .5, n)
    hip.reset_matvec_count()
    fused = hip.gram_matvec(omega, v)
    assert hip.get_dot_count() == (1, 1)
    two = hip.Tdot(omega * hip.dot(v))
    assert np.abs(fused - two).max() <= 1e-11 * np.abs(two).max()
    Xi = np.hstack([np.ones((n, 1)), X - X.mean(axis=0)])
    ref = Xi.T @ (omega * (Xi @ v))
    assert np.abs(fused - ref).max() <= 1e-10 * np.abs(ref).max()
