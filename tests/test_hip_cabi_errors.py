"""Error behaviour of the C ABI (include/bbx.h): every misuse returns a
negative status with a message in bbx_last_error() -- never a crash, never a
silent success.  (The reference's FFI precedent, mkl_matvec.py:17-56, checks
no status at all.)"""
from ctypes import byref, c_int, c_void_p

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

ERR_INVALID = -1


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


@pytest.fixture(scope="module")
def design():
    from bayesbridge_amd import HipSparseDesignMatrix
    X = sp.random(200, 30, density=.2, random_state=0, format='csr')
    return HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True)


def test_null_and_range_checks(design):
    from bayesbridge_amd import _lib
    lib = _lib.load()
    h = design.handle
    n, P = design.shape
    v, out = np.ones(P), np.empty(n)
    assert lib.bbx_design_dot(None, _p(v), _p(out)) == ERR_INVALID
    assert 'NULL' in _lib.last_error()
    assert lib.bbx_design_dot(h, None, _p(out)) == ERR_INVALID
    assert lib.bbx_design_tdot(h, _p(out), None) == ERR_INVALID
    assert lib.bbx_design_destroy(None) == 0          # like free(NULL)
    hh = c_void_p()
    # shape / pointer checks of the constructors
    ip = np.zeros(3, dtype=np.int32)
    assert lib.bbx_design_create_csr(0, 4, 0, _p(ip), None, None, None, 1, 0,
                                     0, byref(hh)) == ERR_INVALID
    assert lib.bbx_design_create_csr(2, 4, 0, None, None, None, None, 1, 0, 0,
                                     byref(hh)) == ERR_INVALID
    assert lib.bbx_design_create_csr(2, 4, 0, _p(ip), None, None, None, 1, 0,
                                     9, byref(hh)) == ERR_INVALID  # format
    assert lib.bbx_design_create_csr(2, 4, 0, _p(ip), None, None, None, 1, 99,
                                     0, byref(hh)) == ERR_INVALID  # device
    assert 'device' in _lib.last_error()
    # an all-empty matrix is legal
    assert lib.bbx_design_create_csr(2, 4, 0, _p(ip), None, None, None, 1, 0,
                                     0, byref(hh)) == 0
    o2, oP = np.empty(2), np.empty(5)
    assert lib.bbx_design_dot(hh, _p(np.arange(5.)), _p(o2)) == 0
    assert np.array_equal(o2, [0., 0.])               # intercept only
    assert lib.bbx_design_tdot(hh, _p(np.array([1., 2.])), _p(oP)) == 0
    assert np.array_equal(oP, [3., 0., 0., 0., 0.])
    assert lib.bbx_design_destroy(hh) == 0


def test_cg_sample_argument_checks(design):
    from bayesbridge_amd import _lib
    lib = _lib.load()
    h = design.handle
    n, P = design.shape
    om, phi, z = np.ones(n), np.ones(P), np.zeros(P)
    x0, sd, coef = np.zeros(P), np.ones(P), np.empty(P)
    e1, e2 = np.zeros(n), np.zeros(P)
    it, info = c_int(), c_int()

    def call(n_unshrunk=1, maxiter=10, r1=e1, r2=e2, omega=om):
        return lib.bbx_cg_sample(h, _p(omega), _p(phi), _p(z), _p(x0), _p(sd),
                                 n_unshrunk, _p(r1), _p(r2), 0, maxiter, 1e-6,
                                 _p(coef), byref(it), byref(info))
    assert call() >= 0
    assert call(n_unshrunk=-1) == ERR_INVALID
    assert call(n_unshrunk=P + 1) == ERR_INVALID
    assert call(maxiter=-3) == ERR_INVALID
    assert call(r1=None) == ERR_INVALID               # only one of the pair
    assert 'randn' in _lib.last_error()
    assert call(omega=None) == ERR_INVALID
    # exhausted iteration budget: positive status = SciPy's info = maxiter
    phi_hard = np.full(P, 1e-3)
    st = lib.bbx_cg_sample(h, _p(om), _p(phi_hard), _p(np.ones(P)), _p(x0),
                           _p(sd), 1, _p(np.ones(n)), _p(np.ones(P)), 0, 2,
                           1e-12, _p(coef), byref(it), byref(info))
    assert st == 2 and info.value == 2 and it.value == 2
    # non-finite input is reported, not propagated silently
    bad = om.copy()
    bad[3] = np.nan
    st = call(omega=bad)
    assert st == -4 and 'non-finite' in _lib.last_error()   # BBX_ERR_NUMERIC


def test_chain_argument_checks(design):
    from bayesbridge_amd import _lib
    lib = _lib.load()
    h = design.handle
    n, P = design.shape
    y, nt = np.zeros(n), np.ones(n)
    sdu = np.array([np.inf])
    c = c_void_p()

    def create(model=_lib.MODEL_LOGIT, nu=1, alpha=.5, outcome=y, sd=sdu):
        return lib.bbx_chain_create(h, model, _p(outcome), _p(nt), nu, _p(sd),
                                    alpha, 1., 0., 0., 1, byref(c))
    assert create(model=7) == ERR_INVALID
    assert create(nu=P + 1) == ERR_INVALID
    assert create(alpha=0.) == ERR_INVALID
    assert create(alpha=2.5) == ERR_INVALID
    assert create(outcome=None) == ERR_INVALID
    assert create(sd=None) == ERR_INVALID
    assert create() == 0
    assert lib.bbx_chain_run(c, 0, 0, 1, 500, 0., None, None, None, None,
                             None, None) == 0             # nothing to do
    assert lib.bbx_chain_run(c, -1, 0, 1, 500, 0., None, None, None, None,
                             None, None) == ERR_INVALID
    assert lib.bbx_chain_run(c, 5, 6, 1, 500, 0., None, None, None, None,
                             None, None) == ERR_INVALID   # burn-in > n_iter
    assert lib.bbx_chain_run(c, 5, 0, 0, 500, 0., None, None, None, None,
                             None, None) == ERR_INVALID   # thin < 1
    assert lib.bbx_chain_set_iteration(c, -1) == ERR_INVALID
    assert lib.bbx_chain_destroy(c) == 0
    assert lib.bbx_chain_get_state(None, None, None, None, None) == ERR_INVALID
