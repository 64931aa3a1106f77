"""Size-independent properties at BASELINE config 3's FULL size (logit,
binary CSR 1 000 000 x 50 000, nnz ~ 1e8), where the CPU oracle needs ~17 s per
Gibbs iteration: adjointness and linearity of the operator, agreement with an
independent device product (torch sparse CSR), and the defining property of a
CG draw -- the returned coefficient solves the perturbed normal equations to
the requested tolerance when the residual is RE-COMPUTED from scratch with
separate operator calls (cg_sampler.py:66-80,104-109)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, P_MAIN, FREQ = 1000000, 50000, .002


@pytest.fixture(scope="module")
def full_design():
    import torch
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    indptr, indices = simulate.simulate_binary_csr_device(
        N, P_MAIN, FREQ, seed=111)
    nnz = indices.numel()
    offset = torch.bincount(indices.long(), minlength=P_MAIN).double() / N
    hip = HipSparseDesignMatrix.from_device_csr(
        N, P_MAIN, nnz, indptr.data_ptr(), indices.data_ptr(), None,
        offset.data_ptr(), add_intercept=True, device=0, storage='tiled')
    torch.cuda.synchronize()
    yield hip, indptr, indices, offset
    del hip


def test_full_size_shape_and_format(full_design):
    hip, indptr, indices, _ = full_design
    assert hip.shape == (N, P_MAIN + 1)
    assert hip.nnz == indices.numel() and abs(hip.nnz - 1e8) < 2e6
    assert hip.storage_format == 'tiled'
    info = hip.tiled_info()
    # one round of workgroups on the 256 CUs, value-free ids
    for side, rows in (('X', N), ('Xt', P_MAIN)):
        n_panel = -(-rows // info[side]['PR'])
        assert n_panel * info[side]['G'] <= 256
    # 2-byte ids + padding + schedules: well under the 4 B/entry of int32 CSR
    dot_bytes, tdot_bytes = hip.matvec_bytes
    assert dot_bytes < 2.6 * hip.nnz and tdot_bytes < 2.8 * hip.nnz


def test_full_size_adjoint_linear_and_independent_product(full_design):
    import torch
    hip, indptr, indices, offset = full_design
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
    # independent product: torch's CSR kernels on the same arrays
    X = torch.sparse_csr_tensor(
        indptr.long(), indices.long(),
        torch.ones(indices.numel(), dtype=torch.float64, device='cuda'),
        size=(n, P - 1))
    v1d = torch.from_numpy(v1).cuda()
    wd = torch.from_numpy(w).cuda()
    ref_dot = v1d[0] + X @ v1d[1:] - torch.dot(offset, v1d[1:])
    assert np.abs(t1 - ref_dot.cpu().numpy()).max() <= 1e-11 * np.abs(t1).max()
    sw = wd.sum()
    ref_t = torch.cat([sw.reshape(1), X.t() @ wd - sw * offset])
    assert np.abs(g - ref_t.cpu().numpy()).max() <= 1e-11 * np.abs(g).max()
    # matvec counters follow the reference's dot_count / Tdot_count
    hip.reset_matvec_count()
    hip.dot(v1), hip.Tdot(w), hip.Tdot(w)
    assert hip.n_matvec == 3


def test_full_size_cg_draw_solves_the_perturbed_system(full_design):
    from bayesbridge_amd import HipCGSampler
    hip, *_ = full_design
    n, P = hip.shape
    rng = np.random.default_rng(6)
    omega = rng.gamma(2., .125, n)                  # Polya-Gamma-like scale
    phi = 1. / (.05 * rng.gamma(1., 1., P) + .01)   # prior_prec_sqrt
    phi[0] = 0.                                     # flat prior on intercept
    y = rng.standard_normal(n)
    z = hip.Tdot(omega * y)
    sd = np.ones(P)
    atol = 1e-5 * np.sqrt(P)                        # reg_coef_sampler.py:95
    sampler = HipCGSampler(n_coef_wo_shrinkage=1)
    seed = 77
    coef, info = sampler.sample(
        hip, omega, phi, z, coef_cg_init=np.zeros(P), coef_scaled_sd=sd,
        maxiter=500, atol=atol, seed=seed)
    assert info['converged'] and info['valid_input']
    assert 3 <= info['n_iter'] < 500
    # rebuild the right-hand side from the same global-stream draws
    np.random.seed(seed)
    eta1, eta2 = np.random.randn(n), np.random.randn(P)
    b = z + hip.Tdot(np.sqrt(omega) * eta1) + phi * eta2
    s = np.empty(P)
    s[0] = 2. * sd[0]                               # cg_sampler.py:128-138
    s[1:] = 1. / phi[1:]
    resid = s * (b - (hip.Tdot(omega * hip.dot(coef)) + phi ** 2 * coef))
    # the recurrence's residual passed ||r|| < atol; the recomputed one may
    # differ by accumulated rounding only
    assert np.linalg.norm(resid) <= 1.05 * atol
    # and the draw is reproducible bit for bit
    coef2, info2 = sampler.sample(
        hip, omega, phi, z, coef_cg_init=np.zeros(P), coef_scaled_sd=sd,
        maxiter=500, atol=atol, seed=seed)
    assert info2['n_iter'] == info['n_iter']
    assert np.array_equal(coef, coef2)


def test_full_size_pair_products_equal_the_single_chain_operator(full_design):
    """K = 2 tiled products (the layout sized for two right-hand sides, one pass
    over the id stream) at full size against the single-chain kernels."""
    from bayesbridge_amd import HipChainBatch, HipGibbsChain
    hip, *_ = full_design
    n, P = hip.shape
    rng = np.random.default_rng(23)
    y = (rng.random(n) < .3).astype(np.float64)
    chains = [HipGibbsChain(hip, 'logit', y, n_trial=np.ones(n),
                            sd_unshrunk=[2.], bridge_exponent=.5, slab_size=2.,
                            seed=s) for s in (1, 2)]
    batch = HipChainBatch(chains, allow_slow=True)
    V, W = rng.standard_normal((2, P)), rng.standard_normal((2, n))
    T, G = batch.dot(V), batch.Tdot(W)
    for c in range(2):
        t, g = hip.dot(V[c]), hip.Tdot(W[c])
        assert np.abs(T[c] - t).max() <= 1e-11 * np.abs(t).max()
        assert np.abs(G[c] - g).max() <= 1e-11 * np.abs(g).max()
