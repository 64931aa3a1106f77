"""GPU: a VALUED sparse design whose value stream alone is past 4 GiB
(3 000 000 x 50 000, ~6e8 stored entries: 4.8 GB of f64 values, 5.6 GB in the
tiled layout's steps) stays in the LDS-tiled format -- the kernels address a
workgroup's stretch of the id and value streams through a 64-bit base and
32-bit offsets inside it (csrc/tiled_layout.hpp BatchDesc, TiledHost::wg_quad0)
-- and agrees with an independent device product (torch sparse CSR).  Before
round 5 such a design fell back to the reference-layout kernels, 12-19x slower
per product.  SciPy's CSR, what SparseDesignMatrix holds
(design_matrix/sparse_matrix.py:49), has no such cliff."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, P_MAIN, FREQ = 3000000, 50000, .004


def test_valued_design_past_the_4gib_value_stream_stays_tiled():
    import torch
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    t0 = time.time()
    indptr, indices = simulate.simulate_binary_csr_device(
        N, P_MAIN, FREQ, seed=7)
    nnz = int(indices.numel())
    assert nnz > 5.4e8 and 8 * nnz > 2 ** 32 and nnz < 2 ** 31
    gen = torch.Generator(device='cuda')
    gen.manual_seed(3)
    vals = torch.rand(nnz, generator=gen, device='cuda',
                      dtype=torch.float64) * 2. + .25
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    hip = HipSparseDesignMatrix.from_device_csr(
        N, P_MAIN, nnz, indptr.data_ptr(), indices.data_ptr(),
        vals.data_ptr(), None, add_intercept=False, device=0, storage='auto')
    t_build = time.time() - t0 - t_gen
    assert hip.storage_format == 'tiled'
    info = hip.tiled_info()
    assert not info['X']['packed'] and not info['Xt']['packed']
    # the value stream of either orientation is past 32-bit byte offsets
    assert info['X']['n_quad'] * 64 * 64 > 2 ** 32
    assert info['Xt']['n_quad'] * 64 * 64 > 2 ** 32
    rng = np.random.default_rng(2)
    v, w = rng.standard_normal(P_MAIN), rng.standard_normal(N)
    t, g = hip.dot(v), hip.Tdot(w)
    X = torch.sparse_csr_tensor(indptr.long(), indices.long(), vals,
                                size=(N, P_MAIN))
    vd, wd = torch.from_numpy(v).cuda(), torch.from_numpy(w).cuda()
    ref_t = (X @ vd).cpu().numpy()
    assert np.abs(t - ref_t).max() <= 1e-11 * np.abs(ref_t).max()
    del ref_t
    # X^T w through index_add (torch's CSC product needs the transposed copy)
    rows = torch.repeat_interleave(
        torch.arange(N, device='cuda'), (indptr[1:] - indptr[:-1]).long())
    ref_g = torch.zeros(P_MAIN, dtype=torch.float64, device='cuda')
    ref_g.index_add_(0, indices.long(), vals * wd[rows])
    ref_g = ref_g.cpu().numpy()
    # (index_add_ adds in no fixed order: rounding of ~12 000-term sums)
    assert np.abs(g - ref_g).max() <= 1e-10 * np.abs(ref_g).max()
    # adjointness on the device products themselves
    lhs, rhs = np.dot(t, w), np.dot(v, g)
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.)
    print("nnz %d: generated in %.0f s, design built in %.0f s, storage "
          "%.1f GB" % (nnz, t_gen, t_build, hip.storage_bytes / 1e9))
