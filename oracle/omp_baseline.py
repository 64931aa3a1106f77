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
        lib.oracle_omp_design_create.restype = c_void_p
        lib.oracle_omp_design_create.argtypes = (
            [c_int64, c_int64, c_int] + [c_void_p] * 7 + [c_int])
        lib.oracle_omp_design_destroy.argtypes = [c_void_p]
        lib.oracle_omp_design_destroy.restype = None
        lib.oracle_omp_dot.argtypes = [c_void_p] * 3
        lib.oracle_omp_tdot.argtypes = [c_void_p] * 3
        lib.oracle_omp_cg_sample.argtypes = (
            [c_void_p] * 6 + [c_int] + [c_void_p] * 2 +
            [c_int, c_double, c_void_p, POINTER(c_int)])
        _lib = lib
    return _lib


def cpu_quota():
    """CPUs' worth of time the cgroup grants this container (cgroup v2
    `cpu.max`, v1 `cpu.cfs_quota_us`), or None when unlimited.  The GPU boxes
    of the pool show 256 cores but grant 16: threads beyond the quota are
    throttled by the scheduler (measured there: 342 / 27 GB/s and 25 / 295 GB/s
    for the two products at 64 and 128 threads, 0.6 GB/s at 256)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return max(1, int(-(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        if quota > 0:
            return max(1, -(-quota // period))
    except (OSError, ValueError):
        pass
    return None


def usable_cores():
    """Cores this process can actually keep busy: the affinity mask (not the
    machine), capped by the container's CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = cpu_quota()
    return min(n, quota) if quota else n


def _p(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class OmpSparseDesign:
    """X~ = [1 | X - 1 offset^T] (sparse_matrix.py:68-129) with both
    orientations stored as value-free (binary designs) CSR inside the C++
    library, each thread's share first-touched by that thread (NUMA); products
    run on `n_threads` cores (default: every core of the affinity mask)."""
    use_cupy = False

    def __init__(self, X, center_predictor=True, add_intercept=True,
                 n_threads=0):
        self.lib = load()
        X = sp.csr_matrix(X)
        X.sort_indices()
        Xt = X.T.tocsr()
        Xt.sort_indices()
        self.n, self.p = X.shape
        self.nnz = X.nnz
        self.intercept = 1 if add_intercept else 0
        self.binary = bool(np.all(X.data == 1.))
        indptr = np.ascontiguousarray(X.indptr, dtype=np.int32)
        indices = np.ascontiguousarray(X.indices, dtype=np.int32)
        data = None if self.binary else np.ascontiguousarray(X.data)
        t_indptr = np.ascontiguousarray(Xt.indptr, dtype=np.int32)
        t_indices = np.ascontiguousarray(Xt.indices, dtype=np.int32)
        t_data = None if self.binary else np.ascontiguousarray(Xt.data)
        self.offset = np.asarray(X.mean(axis=0)).ravel() if center_predictor \
            else np.zeros(self.p)
        self.offset = np.ascontiguousarray(self.offset, dtype=np.float64)
        self.n_threads = int(n_threads) or usable_cores()
        self._h = self.lib.oracle_omp_design_create(
            self.n, self.p, self.intercept, _p(indptr), _p(indices), _p(data),
            _p(t_indptr), _p(t_indices), _p(t_data), _p(self.offset),
            self.n_threads)
        if not self._h:
            raise MemoryError("oracle_omp_design_create failed")

    def __del__(self):
        h = getattr(self, '_h', None)
        if h:
            self.lib.oracle_omp_design_destroy(h)
            self._h = None

    @property
    def shape(self):
        return self.n, self.p + self.intercept

    @property
    def product_bytes(self):
        """(X~ v, X~^T w): bytes one product streams -- SURVEY.md 8(d)'s
        formula for the format actually read (int32 ids, values only when
        stored, row pointers, vector in and out)."""
        per = 4 + (0 if self.binary else 8)
        P = self.p + self.intercept
        return (self.nnz * per + 4 * (self.n + 1) + 8 * (P + self.n),
                self.nnz * per + 4 * (self.p + 1) + 8 * (P + self.n))

    def dot(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        out = np.empty(self.n)
        self.lib.oracle_omp_dot(self._h, _p(v), _p(out))
        return out

    def Tdot(self, w):
        w = np.ascontiguousarray(w, dtype=np.float64)
        out = np.empty(self.p + self.intercept)
        self.lib.oracle_omp_tdot(self._h, _p(w), _p(out))
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
            self._h, _p(omega), _p(phi), _p(z), _p(x0), _p(sd),
            int(n_unshrunk), _p(e1), _p(e2), int(maxiter), float(atol),
            _p(coef), byref(n_iter))
        return coef, {'n_iter': n_iter.value, 'valid_input': info >= 0,
                      'converged': info == 0}
