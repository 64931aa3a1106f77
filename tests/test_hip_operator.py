"""GPU parity of the design operator: libbbx (through the C ABI / ctypes) vs
the CPU oracle on identical seeded inputs.

Mirrors the reference's tests/test_design_matrix.py:12-24,49-61 (dot/Tdot with
intercept + centring equal the explicit matrix, atol = rtol = 1e-5 there) and
adds the edge cases of the domain: empty rows/columns, no intercept, no
centring, non-binary values, skewed column counts.

Tolerance: the reference's own bound is 1e-5; the oracle comparison uses
|diff| <= 1e-11 * scale (f64 sums in a different order)."""
import os

import numpy as np
import pytest
import scipy.sparse as sparse

import oracle
from helpers import mixed_design

pytestmark = pytest.mark.gpu

REF_ATOL = REF_RTOL = 10e-6   # tests/test_design_matrix.py:8-9


def _hip(X, storage, **kw):
    from bayesbridge_amd import HipSparseDesignMatrix
    return HipSparseDesignMatrix(X, storage=storage, **kw)


def _check(hip, ora, seed=0, tol=1e-11):
    n, P = ora.shape
    assert hip.shape == (n, P)
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(P)
    w = rng.standard_normal(n)
    a, b = hip.dot(v), ora.dot(v)
    scale = max(1., np.abs(b).max())
    assert np.abs(a - b).max() <= tol * scale
    a, b = hip.Tdot(w), ora.Tdot(w)
    scale = max(1., np.abs(b).max())
    assert np.abs(a - b).max() <= tol * scale


STORAGES = ['csr', 'tiled']


@pytest.mark.parametrize("storage", STORAGES)
def test_sparse_intercept_and_centering_vs_explicit(storage):
    # tests/test_design_matrix.py:12-24
    X = mixed_design(100, 10, binary_frac=.5, seed=0)
    hip = _hip(X, storage, center_predictor=True, add_intercept=True)
    A = X.toarray()
    A = A - A.mean(axis=0)[None, :]
    A = np.hstack((np.ones((100, 1)), A))
    rng = np.random.default_rng(1)
    w, v = rng.standard_normal(100), rng.standard_normal(11)
    assert np.allclose(hip.dot(v), A.dot(v), atol=REF_ATOL, rtol=REF_RTOL)
    assert np.allclose(hip.Tdot(w), A.T.dot(w), atol=REF_ATOL, rtol=REF_RTOL)
    assert hip.get_dot_count() == (1, 1)
    # the explicit matrix (the reference's sparse toarray is broken upstream,
    # sparse_matrix.py:198-202; this one returns what dot() applies)
    assert np.allclose(hip.toarray(), A, atol=1e-12)


@pytest.mark.parametrize("storage", STORAGES)
@pytest.mark.parametrize("center,intercept", [(True, True), (False, True),
                                              (True, False), (False, False)])
def test_sparse_vs_oracle_flags(storage, center, intercept):
    X = mixed_design(257, 33, binary_frac=.6, seed=2)
    hip = _hip(X, storage, center_predictor=center, add_intercept=intercept)
    ora = oracle.OracleSparseDesign(X, center_predictor=center,
                                    add_intercept=intercept)
    _check(hip, ora, seed=3)


@pytest.mark.parametrize("storage", STORAGES)
def test_binary_skewed_columns(storage):
    # all-ones values => value-free kernels; column counts 1 ... 0.5 n
    X = mixed_design(20000, 700, binary_frac=1., freq=.02, seed=5)
    assert np.all(X.data == 1.)
    hip = _hip(X, storage, center_predictor=True, add_intercept=True)
    ora = oracle.OracleSparseDesign(X, center_predictor=True,
                                    add_intercept=True)
    _check(hip, ora, seed=6)


@pytest.mark.parametrize("storage", STORAGES)
def test_empty_rows_and_ragged(storage):
    rng = np.random.default_rng(7)
    X = sparse.random(513, 70, density=.03, random_state=8, format='lil')
    X[5, :] = 0
    X[100:140, :] = 0            # a band of empty rows
    X[:, 69] = 0                 # trailing empty column (kept: made non-constant below)
    X[3, 69] = 2.5
    X[200, :] = rng.standard_normal(70)   # one full row
    X[:, 7] = rng.standard_normal((513, 1))  # one full column
    X = X.tocsr()
    hip = _hip(X, storage, center_predictor=True, add_intercept=True)
    ora = oracle.OracleSparseDesign(X, center_predictor=True,
                                    add_intercept=True)
    _check(hip, ora, seed=9)


@pytest.mark.parametrize("storage", STORAGES)
def test_tiny_shapes(storage):
    for n, p in [(2, 1), (2, 3), (3, 2), (65, 1), (2, 65), (64, 64)]:
        X = sparse.csr_matrix(
            np.random.default_rng(n * 100 + p).standard_normal((n, p)))
        hip = _hip(X, storage, center_predictor=False, add_intercept=False)
        ora = oracle.OracleSparseDesign(X, center_predictor=False,
                                        add_intercept=False)
        _check(hip, ora, seed=n + p)


@pytest.mark.parametrize("storage", STORAGES)
def test_linearity_and_adjointness_large(storage):
    # size-independent properties at a size the oracle would not enjoy:
    # <X~ v, w> == <v, X~^T w>, and X~(a v1 + v2) == a X~ v1 + X~ v2.
    from bayesbridge_amd import simulate
    X = simulate.simulate_binary_csr_fast(200000, 5000, .004, seed=10)
    hip = _hip(X, storage, center_predictor=True, add_intercept=True)
    n, P = hip.shape
    rng = np.random.default_rng(11)
    v1, v2, w = rng.standard_normal(P), rng.standard_normal(P), \
        rng.standard_normal(n)
    lhs = np.dot(hip.dot(v1), w)
    rhs = np.dot(v1, hip.Tdot(w))
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.)
    lin = hip.dot(2.5 * v1 + v2) - (2.5 * hip.dot(v1) + hip.dot(v2))
    assert np.abs(lin).max() <= 1e-10 * max(1., np.abs(hip.dot(v1)).max())


def test_bad_arguments_raise():
    from bayesbridge_amd import HipSparseDesignMatrix, BbxError
    X = mixed_design(50, 8, seed=12)
    with pytest.raises(NotImplementedError):
        HipSparseDesignMatrix(X, dot_format='csc')   # sparse_matrix.py:31-34
    hip = HipSparseDesignMatrix(X)
    with pytest.raises(ValueError):
        hip.dot(np.zeros(3))
    with pytest.raises(ValueError):
        hip.Tdot(np.zeros(3))
    with pytest.raises(NotImplementedError):
        hip.compute_fisher_info(np.ones(50))


def test_tiled_many_panels_and_valued_entries():
    """More row panels than CUs (several rounds of workgroups, separate
    sum kernel) and non-binary values (the value-carrying kernel), checked
    through adjointness, linearity and a sampled comparison with SciPy."""
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    rng = np.random.default_rng(21)
    X = simulate.simulate_binary_csr_fast(1200000, 300, .02, seed=22)
    X = X.copy()
    X.data = rng.standard_normal(X.nnz) + 2.          # valued entries
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    info = hip.tiled_info()
    n_panel = -(-X.shape[0] // info['X']['PR'])
    assert n_panel * info['X']['G'] > 256
    n, P = hip.shape
    v, w = rng.standard_normal(P), rng.standard_normal(n)
    Xv, Xtw = hip.dot(v), hip.Tdot(w)
    off = np.asarray(X.mean(axis=0)).ravel()
    ref_v = v[0] + X @ v[1:] - off @ v[1:]
    ref_w = np.concatenate(([w.sum()], X.T @ w - w.sum() * off))
    assert np.abs(Xv - ref_v).max() <= 1e-10 * np.abs(ref_v).max()
    assert np.abs(Xtw - ref_w).max() <= 1e-10 * np.abs(ref_w).max()
    lhs, rhs = np.dot(Xv, w), np.dot(v, Xtw)
    assert abs(lhs - rhs) <= 1e-9 * max(abs(lhs), abs(rhs), 1.)


def test_cabi_rejects_malformed_csr():
    """The C ABI validates the CSR structure on the device instead of reading
    out of bounds: wrong indptr ends, column out of range, unsorted rows."""
    from ctypes import byref, c_void_p
    from bayesbridge_amd import _lib
    lib = _lib.load()

    def create(indptr, indices, n=3, p=4):
        indptr = np.asarray(indptr, dtype=np.int32)
        indices = np.asarray(indices, dtype=np.int32)
        h = c_void_p()
        st = lib.bbx_design_create_csr(
            n, p, len(indices), indptr.ctypes.data_as(c_void_p),
            indices.ctypes.data_as(c_void_p), None, None, 1, 0,
            _lib.FORMAT_AUTO, byref(h))
        if st == 0:
            lib.bbx_design_destroy(h)
        return st, _lib.last_error()

    assert create([0, 2, 3, 5], [0, 3, 1, 0, 2])[0] == 0
    st, msg = create([0, 2, 3, 4], [0, 3, 1, 0, 2])
    assert st < 0 and 'indptr' in msg
    st, msg = create([0, 3, 2, 5], [0, 3, 1, 0, 2])
    assert st < 0 and 'indptr' in msg
    st, msg = create([0, 2, 3, 5], [0, 4, 1, 0, 2])
    assert st < 0 and 'out of range' in msg
    st, msg = create([0, 2, 3, 5], [3, 0, 1, 0, 2])
    assert st < 0 and 'ascending' in msg
    # duplicates are legal and add up (SciPy semantics)
    from bayesbridge_amd import HipSparseDesignMatrix
    import scipy.sparse as sp
    X = sp.csr_matrix((np.ones(4), np.array([1, 1, 2, 0], dtype=np.int32),
                       np.array([0, 3, 3, 4], dtype=np.int32)), shape=(3, 3))
    assert not X.has_canonical_format
    for storage in STORAGES:
        hip = HipSparseDesignMatrix(X, add_intercept=False, storage=storage)
        v = np.array([1., 10., 100.])
        np.testing.assert_allclose(hip.dot(v), [120., 0., 1.], rtol=1e-14)
        np.testing.assert_allclose(hip.Tdot(v), [100., 2., 1.], rtol=1e-14)


@pytest.mark.parametrize("case", range(24))
def test_random_shapes_and_patterns(case):
    """Randomised shapes against SciPy: one to several column blocks
    (p around the 16128-column slice width), panels from a handful of rows to
    several thousand, rows from empty to dense, binary and valued entries,
    duplicates, with and without intercept / centring."""
    from bayesbridge_amd import HipSparseDesignMatrix
    from helpers import random_sparse_case
    X, binary, rng = random_sparse_case(case)
    n, p = X.shape
    center, intercept = bool(case & 4), bool(case & 8)
    from bayesbridge_amd.design_matrix import remove_intercept_indicator
    Xr = remove_intercept_indicator(X.copy())
    if Xr.shape[1] == 0:
        pytest.skip("all columns constant")
    for storage in STORAGES:
        hip = HipSparseDesignMatrix(X.copy(), center_predictor=center,
                                    add_intercept=intercept, storage=storage)
        nn, P = hip.shape
        v, w = rng.standard_normal(P), rng.standard_normal(nn)
        off = np.asarray(Xr.mean(axis=0)).ravel() if center \
            else np.zeros(Xr.shape[1])
        a = 1 if intercept else 0
        ref_v = (v[0] if intercept else 0.) + Xr @ v[a:] - off @ v[a:]
        ref_w = Xr.T @ w - w.sum() * off
        if intercept:
            ref_w = np.concatenate(([w.sum()], ref_w))
        tol_v = 1e-11 * max(1., np.abs(ref_v).max())
        tol_w = 1e-11 * max(1., np.abs(ref_w).max())
        assert np.abs(hip.dot(v) - ref_v).max() <= tol_v, (storage, n, p)
        assert np.abs(hip.Tdot(w) - ref_w).max() <= tol_w, (storage, n, p)


@pytest.mark.parametrize("case", range(24))
def test_random_mixed_designs_split_by_value(case):
    """The randomised cases again as MIXED designs -- most entries 1.0, some
    columns with Gaussian values, up to three fully dense continuous columns
    (tests/helper.py:13 builds simulate_design(..., binary_frac=.9)): the
    tiled format stores them split by value (value-free part + dense block +
    valued rest, spmv_tiled.hip HybridParts); products against SciPy, the two
    storages against each other."""
    from bayesbridge_amd import HipSparseDesignMatrix
    from bayesbridge_amd.design_matrix import remove_intercept_indicator
    from helpers import random_sparse_case
    X, _, rng = random_sparse_case(case)
    X = X.tocsr().astype(np.float64)
    X.sum_duplicates()
    n, p = X.shape
    X.data[:] = 1.
    valued_cols = rng.random(p) < .15
    mask = valued_cols[X.indices]
    X.data[mask] = rng.standard_normal(int(mask.sum()))
    n_dense = int(case % 4) if n >= 130 else 0
    if n_dense:
        dense = sparse.csr_matrix(rng.standard_normal((n, n_dense)))
        X = sparse.hstack([X, dense]).tocsr()
    X.sort_indices()
    center, intercept = bool(case & 4), bool(case & 8)
    Xr = remove_intercept_indicator(X.copy())
    if Xr.shape[1] == 0:
        pytest.skip("all columns constant")
    outs = {}
    for storage in STORAGES:
        hip = HipSparseDesignMatrix(X.copy(), center_predictor=center,
                                    add_intercept=intercept, storage=storage)
        nn, P = hip.shape
        if storage == 'tiled':
            info = hip.hybrid_info
            ones = int((Xr.data == 1.).sum())
            if n_dense and ones >= .3 * Xr.nnz:
                assert info is not None and info['dense_cols'] >= 1, info
            if info is not None:
                assert info['ones_nnz'] == ones
                assert info['ones_nnz'] + info['rest_nnz'] + \
                    info['dense_nnz'] == Xr.nnz
        rs = np.random.default_rng(case)
        v, w = rs.standard_normal(P), rs.standard_normal(nn)
        off = np.asarray(Xr.mean(axis=0)).ravel() if center \
            else np.zeros(Xr.shape[1])
        a = 1 if intercept else 0
        ref_v = (v[0] if intercept else 0.) + Xr @ v[a:] - off @ v[a:]
        ref_w = Xr.T @ w - w.sum() * off
        if intercept:
            ref_w = np.concatenate(([w.sum()], ref_w))
        tol_v = 1e-11 * max(1., np.abs(ref_v).max())
        tol_w = 1e-11 * max(1., np.abs(ref_w).max())
        outs[storage] = (hip.dot(v), hip.Tdot(w))
        assert np.abs(outs[storage][0] - ref_v).max() <= tol_v, (storage, n, p)
        assert np.abs(outs[storage][1] - ref_w).max() <= tol_w, (storage, n, p)


def test_mixed_design_of_the_reference_helper_is_stored_split():
    """tests/helper.py:13 -- simulate_design(n, p, binary_frac=.9): 10 % of the
    columns are dense Gaussian.  Split by value the products agree with the
    reference layout's, a CG draw with the oracle's, and a device chain on it
    runs (the golden chain_logit_mixed_initcoef test covers exact parity)."""
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix
    from helpers import cg_inputs, mixed_design
    import oracle
    X = mixed_design(20000, 600, binary_frac=.9, seed=5)
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    info = hip.hybrid_info
    assert info is not None and 55 <= info['dense_cols'] <= 60, info
    assert info['rest_nnz'] == 0
    csr = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='csr')
    assert csr.hybrid_info is None
    n, P = hip.shape
    rng = np.random.default_rng(2)
    v, w = rng.standard_normal(P), rng.standard_normal(n)
    assert np.abs(hip.dot(v) - csr.dot(v)).max() <= 1e-11 * np.abs(csr.dot(v)).max()
    assert np.abs(hip.Tdot(w) - csr.Tdot(w)).max() <= 1e-11 * np.abs(csr.Tdot(w)).max()
    inp = cg_inputs(n, P, seed=4, lam_log_sd=.3)
    ora = oracle.OracleSparseDesign(X, center_predictor=True, add_intercept=True)
    atol = 10e-6 * np.sqrt(P)
    c_o, i_o = oracle.cg_sample(
        ora, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
        inp['coef_cg_init'], inp['coef_scaled_sd'], inp['n_unshrunk'],
        inp['randn_n'], inp['randn_P'], 500, atol)

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
    assert i_h['converged'] and abs(i_h['n_iter'] - i_o['n_iter']) <= 2
    tol = 1e-6 if i_h['n_iter'] == i_o['n_iter'] else 1e-5
    assert np.abs(c_h - c_o).max() <= tol * max(1., np.abs(c_o).max())


@pytest.mark.parametrize("n_dense", [1, 5, 8, 9, 20, 127, 130, 300, 700, 1501,
                                     2100, 4501, 8192])
def test_dense_columns_ride_in_the_dot_epilogue_of_an_operator_application(n_dense):
    """Binary covariates plus a few continuous ones (the OHDSI shape).  Inside
    ONE operator application X~^T (Omega (X~ v)) -- bbx_design_gram_matvec and
    the CG loop -- up to 8 dense columns are handled by the value-free X~ v
    kernel's epilogue (csrc/common.hpp DenseEpi: t += D v_D, and the partials of
    D^T (Omega t) for the transposed product), instead of three kernels of their
    own; 9 ... 1024 take ONE pass over a row-major copy of the dense block
    (hyb_dense_fused_kernel: a wave per row, 1 / 2 / 4 / 8 column pairs per
    lane -- 127, 130, 300 and 700 columns cross those widths; 1025 ... 8192:
    hyb_dense_fused_wg_kernel, a workgroup per row block, 1 / 2 / 4 pairs per
    thread -- 1501, 2100 and 4501 / 8192 (odd widths: the last pair is half used); the reference's generator default,
    simulate_data.py:29 binary_frac=.5, at 10 000 columns is 5 000 of them),
    more stay in the valued layout.  Against the two separate
    products (which never use the fused epilogue) and NumPy; a CG draw against
    the oracle; bitwise repeatable."""
    import scipy.sparse as sparse
    import oracle
    from bayesbridge_amd import HipCGSampler, HipSparseDesignMatrix, simulate
    from helpers import cg_inputs
    n, p = (30000, 2500) if n_dense <= 2500 else (6000, n_dense + 500)
    rng = np.random.default_rng(40 + n_dense)
    Xb = simulate.simulate_binary_csr_fast(n, p, .01, seed=9)
    dense_block = rng.standard_normal((n, n_dense))
    # the last n_dense binary columns are replaced by continuous covariates
    X = sparse.hstack([Xb[:, :p - n_dense],
                       sparse.csr_matrix(dense_block)]).tocsr()
    X.sort_indices()
    p = X.shape[1]
    hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                storage='tiled')
    info = hip.hybrid_info
    assert info is not None and info['dense_cols'] == n_dense
    assert info['rest_nnz'] == 0
    P = p + 1
    v, omega = rng.standard_normal(P), rng.gamma(2., .2, n)
    fused = hip.gram_matvec(omega, v)
    again = hip.gram_matvec(omega, v)
    assert np.array_equal(fused, again)
    two = hip.Tdot(omega * hip.dot(v))          # separate kernels
    assert np.abs(fused - two).max() <= 1e-11 * np.abs(two).max()
    off = np.asarray(X.mean(axis=0)).ravel()
    t = v[0] + X @ v[1:] - off @ v[1:]
    wv = omega * t
    ref = np.concatenate(([wv.sum()], X.T @ wv - wv.sum() * off))
    assert np.abs(fused - ref).max() <= 1e-10 * np.abs(ref).max()
    # a CG draw through the same operator
    inp = cg_inputs(n, P, seed=6, lam_log_sd=.3)
    ora = oracle.OracleSparseDesign(X, center_predictor=True, add_intercept=True)
    atol = 10e-6 * np.sqrt(P)
    c_o, i_o = oracle.cg_sample(
        ora, inp['obs_prec'], inp['prior_prec_sqrt'], inp['z'],
        inp['coef_cg_init'], inp['coef_scaled_sd'], inp['n_unshrunk'],
        inp['randn_n'], inp['randn_P'], 500, atol)

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
    assert i_h['converged']
    assert abs(i_h['n_iter'] - i_o['n_iter']) <= max(2, i_o['n_iter'] // 25)
    tol = 1e-6 if i_h['n_iter'] == i_o['n_iter'] else 1e-5
    assert np.abs(c_h - c_o).max() <= tol * max(1., np.abs(c_o).max())


@pytest.mark.parametrize("shape", [(9000, 20000, .004), (700, 40000, .003),
                                   (20000, 1000, .02)])
def test_kernel_equals_cpu_emulator_bitwise(shape):
    """tiled_spmv_kernel against the CPU emulator of its walk
    (csrc/tiled_layout.cpp::emulate_tiled_spmv, also run without a GPU in
    tests/test_tiled_layout_cpu.py): same layout, same additions in the same
    order, so binary designs agree BIT FOR BIT on both products; valued ones
    to rounding (the GPU contracts a*b + c into fused multiply-adds)."""
    from bayesbridge_amd import HipSparseDesignMatrix, simulate
    from helpers import TiledLayoutCpu
    n, p, f = shape
    layout = TiledLayoutCpu()
    X = simulate.simulate_binary_csr_fast(n, p, f, seed=11)
    rng = np.random.default_rng(3)
    # value-free ids: the builder's choice between four 16-bit ids and a group
    # of five per eight bytes (tiled_layout.hpp packed_slot), and each forced
    # (BBX_TILED_PACK is the builder's diagnostic override)
    for binary, pack in ((True, -1), (True, 0), (True, 1), (False, -1)):
        A = X.copy()
        if not binary:
            A.data = rng.standard_normal(A.nnz)
        At = A.T.tocsr()
        At.sort_indices()
        old = os.environ.pop('BBX_TILED_PACK', None)
        if pack >= 0:
            os.environ['BBX_TILED_PACK'] = str(pack)
        try:
            hip = HipSparseDesignMatrix(A.copy(), center_predictor=False,
                                        add_intercept=False, storage='tiled')
        finally:
            os.environ.pop('BBX_TILED_PACK', None)
            if old is not None:
                os.environ['BBX_TILED_PACK'] = old
        v, w = rng.standard_normal(p), rng.standard_normal(n)
        emu_v, info_x = layout.matvec(A, v, packed=pack)
        emu_w, info_t = layout.matvec(At, w, packed=pack)
        geo = hip.tiled_info()
        for side, info in (('X', info_x), ('Xt', info_t)):
            for key in ('W', 'n_block', 'PR', 'G', 'n_quad', 'n_slice',
                        'packed'):
                assert geo[side][key] == info[key], (side, key)
            if pack >= 0:
                assert geo[side]['packed'] == (binary and pack == 1)
            if not binary:
                assert not geo[side]['packed']
        got_v, got_w = hip.dot(v), hip.Tdot(w)
        if binary:
            assert np.array_equal(got_v, emu_v), pack
            assert np.array_equal(got_w, emu_w), pack
        else:
            assert np.abs(got_v - emu_v).max() <= 1e-13 * np.abs(emu_v).max()
            assert np.abs(got_w - emu_w).max() <= 1e-13 * np.abs(emu_w).max()


def test_packed_groups_edge_cases_on_the_device():
    """The rows of tests/test_tiled_layout_cpu.py::packed_edge_case_matrix --
    columns on both sides of the LDS slice's zero slots, gaps of exactly 4095
    and 4096 slots, duplicate column entries, groups of 1 ... 5 entries -- through
    tiled_spmv_kernel<..., PACK> (groups forced: the builder would not choose
    them for a matrix this small), with integer vectors: exact sums."""
    from bayesbridge_amd import HipSparseDesignMatrix
    from test_tiled_layout_cpu import packed_edge_case_matrix
    A = packed_edge_case_matrix()
    rng = np.random.default_rng(8)
    v = rng.integers(-40, 40, A.shape[1]).astype(np.float64)
    w = rng.integers(-40, 40, A.shape[0]).astype(np.float64)
    old = os.environ.pop('BBX_TILED_PACK', None)
    os.environ['BBX_TILED_PACK'] = '1'
    try:
        hip = HipSparseDesignMatrix(A.copy(), center_predictor=False,
                                    add_intercept=False, storage='tiled')
    finally:
        os.environ.pop('BBX_TILED_PACK', None)
        if old is not None:
            os.environ['BBX_TILED_PACK'] = old
    info = hip.tiled_info()
    assert info['X']['packed'] and info['Xt']['packed']
    assert np.array_equal(hip.dot(v), A @ v)
    assert np.array_equal(hip.Tdot(w), A.T @ w)
    # with intercept and centring the input of the product is v + 1 doubles
    # into the caller's array: the 8-byte slice fill (WIDE = false)
    os.environ['BBX_TILED_PACK'] = '1'
    try:
        hipc = HipSparseDesignMatrix(A.copy(), center_predictor=True,
                                     add_intercept=True, storage='tiled')
    finally:
        os.environ.pop('BBX_TILED_PACK', None)
        if old is not None:
            os.environ['BBX_TILED_PACK'] = old
    n, P = hipc.shape
    v1 = rng.standard_normal(P)
    off = np.asarray(A.mean(axis=0)).ravel()
    ref = v1[0] + A @ v1[1:] - off @ v1[1:]
    assert np.abs(hipc.dot(v1) - ref).max() <= 1e-11 * max(1., np.abs(ref).max())


def test_operator_application_gives_the_same_bits_on_every_launch():
    """scripts/soak_products.py: the same operator application launched 3000
    times back to back -- with a second stream running bursts of unrelated
    kernels beside it -- returns the same bits every time, on the headline's
    kind of design (value-free tiled layout, packed or plain ids) and on a mixed
    one (dense block in one pass).  The tiled kernels keep a hand-counted ring
    of asm-issued loads; a hazard there is silent and rare, not loud.
    (10 000 launches each at full size: profiles/r05_soak.txt.)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), 'scripts', 'soak_products.py')
    for what in ('config2', 'mixed_small'):
        out = subprocess.run([sys.executable, script, what, '3000', 'perturb'],
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-500:] + out.stderr[-500:]
        assert '0 of 3000 operator applications differ' in out.stdout
