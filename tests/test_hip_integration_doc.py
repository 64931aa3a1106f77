"""GPU: executes the reference-side ctypes binding printed in INTEGRATION.md
(section 1, the `mkl_matvec.py` analogue a maintainer would add) so that the
documented stub cannot rot: the code block is extracted from the document and
run as is against the golden operator case and one recorded CG call of the
reference."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sparse

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stub():
    from bayesbridge_amd import _lib
    _lib.load()      # torch's HIP runtime first (DESIGN.md "One HIP runtime")
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    code = next(b for b in blocks if "hip_matvec.py" in b)
    # the only substitution: where this checkout keeps the library
    assert 'LoadLibrary("libbbx.so")' in code
    code = code.replace('"libbbx.so"', repr(_lib.LIB_PATH))
    ns = {}
    exec(compile(code, "INTEGRATION.md:hip_matvec.py", "exec"), ns)
    return ns


def test_documented_binding_reproduces_the_golden_operator(stub, golden_dir):
    g = np.load(os.path.join(golden_dir, 'operator_sparse_100x10.npz'))
    X = sparse.csr_matrix(g['X'])
    offset = np.ascontiguousarray(np.asarray(X.mean(axis=0)).ravel())
    h = stub['hip_design_create'](X, offset, 1)
    n, p = X.shape
    dot = stub['hip_dot'](h, g['v'], n)
    tdot = stub['hip_tdot'](h, g['w'], p + 1)
    assert np.abs(dot - g['dot']).max() <= 1e-12
    assert np.abs(tdot - g['Tdot']).max() <= 1e-12
    stub['bbx'].bbx_design_destroy(h)


def test_documented_binding_replays_a_recorded_cg_call(stub, golden_dir):
    g = np.load(os.path.join(golden_dir, 'chain_logit_sparse_cg.npz'))
    X = sparse.csr_matrix(g['X'])
    offset = np.ascontiguousarray(np.asarray(X.mean(axis=0)).ravel())
    h = stub['hip_design_create'](X, offset, 1)
    for it in (0, 4, 9):
        arrays = [np.ascontiguousarray(g['cg_' + k][it]) for k in (
            'obs_prec', 'prior_prec_sqrt', 'z', 'coef_cg_init',
            'coef_scaled_sd')]
        coef, n_iter, info = stub['hip_cg_sample'](
            h, *arrays, int(g['cg_n_unshrunk'][it]),
            np.ascontiguousarray(g['cg_randn_n'][it]),
            np.ascontiguousarray(g['cg_randn_P'][it]),
            int(g['cg_maxiter'][it]), float(g['cg_atol'][it]))
        assert info == 0
        assert abs(n_iter - int(g['cg_n_iter'][it])) <= 2
        tol = 1e-6 if n_iter == int(g['cg_n_iter'][it]) else 1e-5
        assert np.abs(coef - g['cg_coef'][it]).max() <= tol
    stub['bbx'].bbx_design_destroy(h)
