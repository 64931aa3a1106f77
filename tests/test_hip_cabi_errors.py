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


def test_round2_entry_points_argument_checks(design):
    """The entry points added in round 2: gram_matvec, timed / fused bytes,
    device, chain seed / gscale mode / eta / logp, device normals."""
    from ctypes import c_double, c_int64, c_uint64
    from bayesbridge_amd import _lib
    lib = _lib.load()
    h = design.handle
    n, P = design.shape
    om, v, out = np.ones(n), np.ones(P), np.empty(P)
    assert lib.bbx_design_gram_matvec(None, _p(om), _p(v), _p(out)) == ERR_INVALID
    assert lib.bbx_design_gram_matvec(h, None, _p(v), _p(out)) == ERR_INVALID
    assert lib.bbx_design_gram_matvec(h, _p(om), _p(v), None) == ERR_INVALID
    assert lib.bbx_design_gram_matvec(h, _p(om), _p(v), _p(out)) == 0
    assert np.allclose(out, design.Tdot(design.dot(v)), rtol=1e-12, atol=1e-10)
    a, b = c_int64(), c_int64()
    assert lib.bbx_design_timed_bytes(None, byref(a), byref(b)) == ERR_INVALID
    assert lib.bbx_design_timed_bytes(h, byref(a), byref(b)) == 0
    wa, wb = c_int64(), c_int64()
    assert lib.bbx_design_matvec_bytes(h, byref(wa), byref(wb)) == 0
    assert 0 < a.value <= wa.value and 0 < b.value <= wb.value
    assert lib.bbx_design_fused_operator_bytes(h, byref(a)) == 0
    assert a.value == 0                         # sparse: no single-pass kernel
    dev = c_int(-7)
    assert lib.bbx_design_device(None, byref(dev)) == ERR_INVALID
    assert lib.bbx_design_device(h, byref(dev)) == 0 and dev.value == 0
    # which = 2 is the whole operator application; 3 does not exist
    cnt, ms = c_int64(), c_double()
    assert lib.bbx_design_get_timing(h, 2, byref(cnt), byref(ms)) == 0
    assert lib.bbx_design_get_timing(h, 3, byref(cnt), byref(ms)) == ERR_INVALID

    y, nt, sdu = np.zeros(n), np.ones(n), np.array([np.inf])
    c = c_void_p()
    assert lib.bbx_chain_create(h, _lib.MODEL_LOGIT, _p(y), _p(nt), 1, _p(sdu),
                                .5, 1., 0., 0., 77, byref(c)) == 0
    seed = c_uint64()
    assert lib.bbx_chain_get_seed(c, byref(seed)) == 0 and seed.value == 77
    assert lib.bbx_chain_set_seed(c, 78) == 0
    assert lib.bbx_chain_get_seed(c, byref(seed)) == 0 and seed.value == 78
    assert lib.bbx_chain_set_seed(None, 1) == ERR_INVALID
    assert lib.bbx_chain_set_gscale_update(c, 3) == ERR_INVALID
    assert 'mode' in _lib.last_error()
    assert lib.bbx_chain_set_gscale_update(c, _lib.GSCALE_FIXED) == 0
    e1, e2 = np.empty(n), np.empty(P)
    assert lib.bbx_chain_eta(c, -1, _p(e1), _p(e2)) == ERR_INVALID
    assert lib.bbx_chain_eta(None, 0, _p(e1), _p(e2)) == ERR_INVALID
    assert lib.bbx_chain_eta(c, 0, _p(e1), None) == 0     # either may be NULL
    assert lib.bbx_chain_eta(c, 0, None, _p(e2)) == 0
    assert np.all(np.isfinite(e1)) and np.all(np.isfinite(e2))
    ll, lp = c_double(), c_double()
    assert lib.bbx_chain_get_logp(None, byref(ll), byref(lp)) == ERR_INVALID
    assert lib.bbx_chain_get_logp(c, byref(ll), None) == 0
    assert lib.bbx_chain_destroy(c) == 0
    z = np.empty(8)
    assert lib.bbx_device_normal(0, 1, 1, -1, _p(z)) == ERR_INVALID
    assert lib.bbx_device_normal(9, 1, 1, 8, _p(z)) == ERR_INVALID   # device
    assert lib.bbx_device_normal(0, 1, 1, 8, None) == ERR_INVALID
    assert lib.bbx_device_normal(0, 1, 1, 0, None) == 0              # nothing
    assert lib.bbx_device_normal(0, 1, 1, 8, _p(z)) == 0
    assert np.all(np.isfinite(z)) and len(np.unique(z)) == 8


def test_auto_format_falls_back_to_the_reference_layout():
    """BBX_FORMAT_AUTO = "tiled when it applies, else csr" (AUTO
    propagated the tiled builder's failure).  The builder's refusal is forced
    here by an override the layout cannot honour (a subprocess: the builder
    reads its environment once)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = """
import sys
sys.path.insert(0, %r)
import numpy as np, scipy.sparse as sp
from bayesbridge_amd import HipSparseDesignMatrix, BbxError
X = sp.random(300, 40000, density=.01, random_state=1, format='csr')
auto = HipSparseDesignMatrix(X, storage='auto')
assert auto.storage_format == 'csr', auto.storage_format
ref = HipSparseDesignMatrix(X, storage='csr')
v = np.linspace(-1., 1., auto.shape[1])
assert auto.shape == ref.shape
assert np.array_equal(auto.dot(v), ref.dot(v))
try:
    HipSparseDesignMatrix(X, storage='tiled')
except BbxError as e:
    assert 'LDS' in str(e)
    print('FALLBACK_OK')
""" % os.path.join(ROOT, 'bayes-bridge_amd')
    # 8192-row panels next to a 16000-column slice do not fit the CU's LDS
    env = dict(os.environ, BBX_TILED_PR='8192')
    res = subprocess.run([sys.executable, '-c', code], env=env,
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout + res.stderr
    assert 'FALLBACK_OK' in res.stdout
