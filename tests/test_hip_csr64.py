"""GPU: designs handed over with 64-bit index arrays (bbx_design_create_csr64).

SciPy's CSR -- what SparseDesignMatrix holds, design_matrix/sparse_matrix.py:49
-- switches indptr / indices to int64 once a matrix has 2^31 or more stored
entries (scipy.sparse get_index_dtype); the reference's dot / Tdot
(sparse_matrix.py:68-129) work on whatever SciPy holds.  Here:
  * int64 arrays below 2^31 entries: same design, bit for bit, as the int32
    constructor (the library narrows copies);
  * the HOST path (validation, all-ones test, transposition on the host; tiled
    layout only) forced on small designs through BBX_CSR64_BIG_MIN: same
    products bit for bit as the device path -- binary, valued and mixed
    designs, centred, with intercept -- and the same CG draw;
  * a real design of 2.25e9 entries (ten stacked copies of a 2.25e8-entry
    block A, so that X v = tile(A v) and X^T w = A^T sum_b w_b) against the
    products of A's own design."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sparse

from helpers import cg_inputs, mixed_design, random_sparse_case

pytestmark = pytest.mark.gpu


def _pair(X, storage='tiled', **kw):
    """(design from int32 arrays, design from int64 arrays) of one matrix."""
    from bayesbridge_amd import HipSparseDesignMatrix
    X = sparse.csr_matrix(X)
    X.sort_indices()
    n, p = X.shape
    offset = kw.pop('column_offset', None)
    narrow = HipSparseDesignMatrix.from_csr_arrays(
        (n, p), X.indptr.astype(np.int32), X.indices.astype(np.int32), X.data,
        column_offset=offset, storage=storage, **kw)
    wide = HipSparseDesignMatrix.from_csr_arrays(
        (n, p), X.indptr.astype(np.int64), X.indices.astype(np.int64), X.data,
        column_offset=offset, storage=storage, **kw)
    return narrow, wide


def _same_products(a, b, seed=0):
    n, P = a.shape
    assert b.shape == (n, P) and a.nnz == b.nnz
    rng = np.random.default_rng(seed)
    for _ in range(2):
        v, w = rng.standard_normal(P), rng.standard_normal(n)
        assert np.array_equal(a.dot(v), b.dot(v))
        assert np.array_equal(a.Tdot(w), b.Tdot(w))


@pytest.mark.parametrize("storage", ['csr', 'tiled', 'auto'])
def test_int64_index_arrays_below_2_31_entries_equal_the_int32_constructor(storage):
    X, _, _ = random_sparse_case(4)
    narrow, wide = _pair(X, storage=storage)
    assert narrow.storage_format == wide.storage_format
    _same_products(narrow, wide)
    X = mixed_design(3000, 60, binary_frac=.7, seed=2)
    offset = np.asarray(X.mean(axis=0)).ravel()
    narrow, wide = _pair(X, storage=storage, column_offset=offset)
    _same_products(narrow, wide, seed=1)


@pytest.fixture
def host_path(monkeypatch):
    """Every csr64 design with at least one entry takes the >= 2^31 path."""
    monkeypatch.setenv('BBX_CSR64_BIG_MIN', '1')
    yield
    monkeypatch.delenv('BBX_CSR64_BIG_MIN', raising=False)


@pytest.mark.parametrize("case", [1, 2, 4, 7, 9, 12, 15, 20])
def test_host_path_equals_device_path_bitwise_on_random_patterns(case, host_path):
    """Validation + all-ones test + stable transposition on the host must hand
    the layout builders exactly what the device path (radix sort) does: the
    tiled products agree bit for bit, duplicates and empty rows included."""
    X, binary, _ = random_sparse_case(case)
    narrow, wide = _pair(X, storage='tiled', add_intercept=bool(case % 2))
    assert wide.storage_format == 'tiled'
    assert narrow.tiled_info() == wide.tiled_info()
    _same_products(narrow, wide, seed=case)


def test_host_path_on_mixed_centred_designs_and_a_cg_draw(host_path):
    """Mixed binary + continuous columns through the host path: the split into
    value-free part, dense block and valued rest (csrc/spmv_tiled.hip
    build_hybrid) reads the host CSR; centred, with intercept; one draw of the
    CG sampler (cg_sampler.py:20-94) on both designs is the same draw."""
    from bayesbridge_amd import HipCGSampler
    for frac, p in ((.9, 400), (.5, 60)):
        X = mixed_design(6000, p, binary_frac=frac, seed=3)
        offset = np.asarray(X.mean(axis=0)).ravel()
        narrow, wide = _pair(X, storage='tiled', column_offset=offset)
        assert narrow.tiled_info() == wide.tiled_info()
        _same_products(narrow, wide, seed=5)
        n, P = narrow.shape
        inp = cg_inputs(n, P, seed=8)
        kw = dict(coef_cg_init=inp['coef_cg_init'],
                  coef_scaled_sd=inp['coef_scaled_sd'], maxiter=400,
                  atol=1e-6 * np.sqrt(P))
        draws = []
        for d in (narrow, wide):
            np.random.seed(13)
            coef, info = HipCGSampler(1).sample(
                d, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'], **kw)
            assert info['converged']
            draws.append((coef, info['n_iter']))
        assert draws[0][1] == draws[1][1]
        assert np.array_equal(draws[0][0], draws[1][0])


def test_host_path_argument_checks(host_path):
    """Same refusals, same words as the device validation (csrc/api.hip
    validate_csr); the reference-layout format and batch layouts do not exist
    for such a design."""
    from bayesbridge_amd import (BbxError, HipChainBatch, HipGibbsChain,
                                 HipSparseDesignMatrix)
    X = sparse.csr_matrix(np.array([[1., 0, 1, 0], [0, 1, 0, 1],
                                    [1, 0, 0, 1]] * 40))
    ip, ix = X.indptr.astype(np.int64), X.indices.astype(np.int64)

    def create(indptr=ip, indices=ix, storage='auto', shape=X.shape):
        return HipSparseDesignMatrix.from_csr_arrays(shape, indptr, indices,
                                                     storage=storage)
    bad = ip.copy()
    bad[3] = bad[2] - 1
    with pytest.raises(BbxError, match='non-decreasing'):
        create(indptr=bad)
    bad = ix.copy()
    bad[5] = 4
    with pytest.raises(BbxError, match='out of range'):
        create(indices=bad)
    bad = ix.copy()
    bad[0], bad[1] = bad[1], bad[0]
    with pytest.raises(BbxError, match='ascending'):
        create(indices=bad)
    with pytest.raises(BbxError, match='tiled format'):
        create(storage='csr')
    hip = create()
    assert hip.storage_format == 'tiled' and hip.nnz == X.nnz
    assert np.allclose(hip.dot(np.arange(5.)),
                       np.arange(5.)[0] + X @ np.arange(1., 5.))
    y = np.arange(X.shape[0]) % 2
    pair = [HipGibbsChain(hip, 'logit', y, seed=s) for s in (1, 2)]
    with pytest.raises(BbxError, match='one chain at a time'):
        HipChainBatch(pair, allow_slow=True)
    # a single chain runs
    out, _ = pair[0].run(2)
    assert np.all(np.isfinite(out['coef'])) and np.all(out['n_cg_iter'] > 0)


def test_int64_arrays_that_do_not_fit_are_refused_not_wrapped():
    """Narrowing below 2^31 entries must not wrap a column id of 2^32 + 1 into
    a valid one."""
    from bayesbridge_amd import BbxError, HipSparseDesignMatrix
    ip = np.array([0, 2, 3], dtype=np.int64)
    ix = np.array([0, 2 ** 32 + 1, 1], dtype=np.int64)
    with pytest.raises(BbxError, match='out of range'):
        HipSparseDesignMatrix.from_csr_arrays((2, 3), ip, ix)
    ip = np.array([0, 2 ** 32 + 2, 3], dtype=np.int64)
    ix = np.array([0, 1, 1], dtype=np.int64)
    with pytest.raises(BbxError, match='non-decreasing'):
        HipSparseDesignMatrix.from_csr_arrays((2, 3), ip, ix)


def _usable_host_gb():
    import psutil
    gb = psutil.virtual_memory().available / 2 ** 30
    try:
        with open('/sys/fs/cgroup/memory.max') as f:
            cap = f.read().strip()
        with open('/sys/fs/cgroup/memory.current') as f:
            used = int(f.read().strip())
        if cap != 'max':
            gb = min(gb, (int(cap) - used) / 2 ** 30)
    except OSError:
        pass
    return gb


N_A, P_BIG, PER_ROW, COPIES = 250000, 20000, 900, 10


def test_a_design_of_more_than_2_31_stored_entries():
    """2.25e9 stored ones (n = 2.5M, p = 20k): SciPy would hold int64 index
    arrays; the int32 constructor refuses the size, the 64-bit one stores the
    design in the tiled layout (3.6 GB of packed ids per orientation instead of
    18 GB of int64 column ids) and its products are those of the repeated
    block."""
    if _usable_host_gb() < 70:
        pytest.skip("needs ~50 GB of host memory for the index arrays and "
                    "the layout build")
    from bayesbridge_amd import BbxError, HipSparseDesignMatrix
    rng = np.random.default_rng(1)
    t0 = time.time()
    # rows of A: 900 ascending columns, gaps 1..22 (the last one < 20 000)
    gaps = rng.integers(1, 23, size=(N_A, PER_ROW), dtype=np.int32)
    gaps[:, 0] = rng.integers(0, 180, size=N_A)
    cols_a = np.cumsum(gaps, axis=1, dtype=np.int32)
    del gaps
    assert cols_a.max() < P_BIG
    nnz_a = N_A * PER_ROW
    ptr_a = np.arange(N_A + 1, dtype=np.int64) * PER_ROW
    block = HipSparseDesignMatrix.from_csr_arrays(
        (N_A, P_BIG), ptr_a.astype(np.int32), cols_a.ravel(),
        add_intercept=False, storage='tiled')
    import scipy.sparse as sparse
    A_block = sparse.csr_matrix(
        (np.ones(nnz_a), cols_a.ravel().copy(), ptr_a.astype(np.int32)),
        shape=(N_A, P_BIG))
    n, nnz = N_A * COPIES, nnz_a * COPIES
    assert nnz >= 2 ** 31
    indices = np.empty(nnz, dtype=np.int64)
    for b in range(COPIES):
        indices[b * nnz_a:(b + 1) * nnz_a] = cols_a.ravel()
    del cols_a
    indptr = np.arange(n + 1, dtype=np.int64) * PER_ROW
    t_gen = time.time() - t0
    # the int32 constructor says where to go
    from ctypes import byref, c_void_p
    from bayesbridge_amd import _lib
    lib, h, small = _lib.load(), c_void_p(), np.zeros(8, dtype=np.int32)
    assert lib.bbx_design_create_csr(
        n, P_BIG, nnz, small.ctypes.data, small.ctypes.data, None, None, 0, 0,
        0, byref(h)) < 0
    assert b'create_csr64' in lib.bbx_last_error()
    t0 = time.time()
    hip = HipSparseDesignMatrix.from_csr_arrays(
        (n, P_BIG), indptr, indices, add_intercept=False)
    t_build = time.time() - t0
    del indices
    assert hip.storage_format == 'tiled' and hip.nnz == nnz
    assert hip.shape == (n, P_BIG)
    v, w = rng.standard_normal(P_BIG), rng.standard_normal(n)
    t, g = hip.dot(v), hip.Tdot(w)
    # the oracle: SciPy's CSR products of the block (sparse_matrix.py:96,126)
    # -- the design is COPIES stacked copies of it, so X v is the block's
    # product tiled and X^T w the block's transposed product of the summed w
    ref_t = np.tile(A_block.dot(v), COPIES)
    assert np.abs(t - ref_t).max() <= 1e-11 * np.abs(ref_t).max()
    ref_g = A_block.T.dot(w.reshape(COPIES, N_A).sum(axis=0))
    assert np.abs(g - ref_g).max() <= 1e-10 * np.abs(ref_g).max()
    # ... and the HIP operator of the block alone agrees with both
    assert np.abs(np.tile(block.dot(v), COPIES) - ref_t).max() \
        <= 1e-11 * np.abs(ref_t).max()
    lhs, rhs = np.dot(t, w), np.dot(v, g)
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.)
    # the data part of the CG operator through the launches the CG loop uses
    omega = rng.random(n) + .5
    gm = hip.gram_matvec(omega, v)
    ref_gm = A_block.T.dot((omega * ref_t).reshape(COPIES, N_A).sum(axis=0))
    assert np.abs(gm - ref_gm).max() <= 1e-9 * np.abs(ref_gm).max()
    # a device chain runs on it: three Gibbs iterations of the logit model
    from bayesbridge_amd import HipGibbsChain
    y = (rng.random(n) < .3).astype(np.float64)
    chain = HipGibbsChain(hip, 'logit', y, sd_unshrunk=[], slab_size=1.,
                          seed=3)
    chain.set_state(np.zeros(P_BIG), None, np.ones(P_BIG), .01)
    chain.init_obs_prec()
    out, n_unconverged = chain.run(3)
    assert n_unconverged == 0 and np.all(out['n_cg_iter'] > 0)
    assert np.all(np.isfinite(out['coef'])) and np.all(np.isfinite(out['logp']))
    n_cg = out['n_cg_iter']
    del chain
    # the rate of the two products at this size (kernel stamps)
    hip.set_timing(True)
    for _ in range(5):
        hip.gram_matvec(omega, v)
    tm, by = hip.get_timing(), hip.timed_bytes
    rate = {k: by[i] / (tm[k][1] / tm[k][0] * 1e-3) / 1e9
            for i, k in enumerate(('dot', 'tdot'))}
    hip.set_timing(False)
    assert min(rate.values()) > 1000.        # GB/s: streaming, not crawling
    print("nnz %d: arrays in %.0f s, design built in %.0f s, storage %.1f GB, "
          "tiled %s; chain n_cg %s; X~ v %.2f ms = %.0f GB/s, X~^T w %.2f ms = "
          "%.0f GB/s"
          % (nnz, t_gen, t_build, hip.storage_bytes / 1e9,
             {k: (d['W'], d['PR'], d['G'], d['packed'])
              for k, d in hip.tiled_info().items()}, n_cg.astype(int).tolist(),
             tm['dot'][1] / tm['dot'][0], rate['dot'],
             tm['tdot'][1] / tm['tdot'][0], rate['tdot']))
