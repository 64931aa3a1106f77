"""Multi-core CPU baseline (TEST INFRASTRUCTURE): ctypes front of
oracle/csrc/oracle_cg_omp.cpp -- the design operator and one CG draw of
cg_sampler.py:20-94 as OpenMP loops.  Used by bench.py's
`cpu_baseline.port_omp` leg and checked against the NumPy oracle in
tests/test_oracle_omp_baseline.py.  Never imported by the product."""
import ctypes
import os
import subprocess
from ctypes import POINTER, byref, c_double, c_int, c_int64, c_void_p

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle_omp.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            subprocess.check_call(["make", "-C", _HERE],
                                  stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(_LIB_PATH)
        lib.oracle_omp_max_threads.restype = c_int
        lib.oracle_omp_dot.argtypes = [
            c_int64, c_int64, c_int] + [c_void_p] * 6 + [c_int]
        lib.oracle_omp_tdot.argtypes = [
            c_int64, c_int64, c_int] + [c_void_p] * 6 + [c_int]
        lib.oracle_omp_cg_sample.argtypes = (
            [c_int64, c_int64, c_int] + [c_void_p] * 12 + [c_int]
            + [c_void_p] * 2 + [c_int, c_double, c_void_p, POINTER(c_int),
                                c_int])
        _lib = lib
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class OmpSparseDesign:
    """X~ = [1 | X - 1 offset^T] (sparse_matrix.py:68-129) with both
    orientations stored as CSR; products run on `n_threads` cores."""
    use_cupy = False

    def __init__(self, X, center_predictor=True, add_intercept=True,
                 n_threads=0):
        self.lib = load()
        X = sp.csr_matrix(X)
        X.sort_indices()
        Xt = X.T.tocsr()
        Xt.sort_indices()
        self.n, self.p = X.shape
        self.intercept = 1 if add_intercept else 0
        binary = bool(np.all(X.data == 1.))
        self.indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
        self.indices = np.ascontiguousarray(X.indices, dtype=np.int32)
        self.data = None if binary else np.ascontiguousarray(X.data)
        self.t_indptr = np.ascontiguousarray(Xt.indptr, dtype=np.int32)
        self.t_indices = np.ascontiguousarray(Xt.indices, dtype=np.int32)
        self.t_data = None if binary else np.ascontiguousarray(Xt.data)
        self.offset = np.asarray(X.mean(axis=0)).ravel() if center_predictor \
            else np.zeros(self.p)
        self.n_threads = int(n_threads) or self.lib.oracle_omp_max_threads()

    @property
    def shape(self):
        return self.n, self.p + self.intercept

    def dot(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.empty(self.n)
        self.lib.oracle_omp_dot(self.n, self.p, self.intercept,
                                _p(self.indptr), _p(self.indices),
                                _p(self.data), _p(self.offset), _p(v), _p(out),
                                self.n_threads)
        return out

    def Tdot(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.empty(self.p + self.intercept)
        self.lib.oracle_omp_tdot(self.n, self.p, self.intercept,
                                 _p(self.t_indptr), _p(self.t_indices),
                                 _p(self.t_data), _p(self.offset), _p(w),
                                 _p(out), self.n_threads)
        return out

    def cg_sample(self, obs_prec, prior_prec_sqrt, z, coef_cg_init,
                  coef_scaled_sd, n_unshrunk, randn_n, randn_P, maxiter, atol):
        """Same contract as oracle.cg_sample."""
        P = self.p + self.intercept
        args = [np.ascontiguousarray(a, dtype=np.float64) for a in (
            np.broadcast_to(obs_prec, (self.n,)), prior_prec_sqrt, z,
            coef_cg_init, coef_scaled_sd, randn_n, randn_P)]
        omega, phi, z, x0, sd, e1, e2 = args
        coef = np.empty(P)
        n_iter = c_int(0)
        info = self.lib.oracle_omp_cg_sample(
            self.n, self.p, self.intercept, _p(self.indptr), _p(self.indices),
            _p(self.data), _p(self.t_indptr), _p(self.t_indices),
            _p(self.t_data), _p(self.offset), _p(omega), _p(phi), _p(z),
            _p(x0), _p(sd), int(n_unshrunk), _p(e1), _p(e2), int(maxiter),
            float(atol), _p(coef), byref(n_iter), self.n_threads)
        return coef, {'n_iter': n_iter.value, 'valid_input': info >= 0,
                      'converged': info == 0}
