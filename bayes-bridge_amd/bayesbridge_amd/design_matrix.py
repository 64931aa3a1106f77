"""Design-matrix operators backed by libbbx.so.

Mirror of the reference's duck-typed operator interface
(design_matrix/abstract_matrix.py:14-77, sparse_matrix.py:18-129,
dense_matrix.py:7-52): same attribute names (`use_cupy`, `intercept_added`,
`centered`), same properties (`shape`, `is_sparse`, `nnz`, `n_matvec`) and the
same methods (`dot`, `Tdot`, `get_dot_count`, `reset_matvec_count`,
`memoize_dot`), so an object of this module drops in wherever the reference
uses SparseDesignMatrix / DenseDesignMatrix on the 'cg' path.  The new
dispatch attribute is `use_hip` (the reference's seam is `use_cupy`,
sparse_matrix.py:35).
"""
import ctypes
import warnings
from ctypes import byref, c_double, c_int, c_int64, c_void_p

import numpy as np
import scipy.sparse as sparse

from . import _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def remove_intercept_indicator(X):
    """Columns that are constant over the rows (an intercept the caller added
    by hand, or numerically the same thing) are dropped: the operator adds its
    own intercept and a constant column would be collinear with it.  Same rule
    as the reference (abstract_matrix.py:93-107): population variance below
    n_rows * 2**-52."""
    n_rows = X.shape[0]
    if sparse.issparse(X):
        first = np.asarray(X.mean(axis=0)).ravel()
        second = np.asarray(X.multiply(X).mean(axis=0)).ravel()
        variance = second - first ** 2
    else:
        variance = np.var(X, axis=0)
    constant = variance < n_rows * 2. ** -52
    if constant.any():
        warnings.warn(
            "%d constant column(s) found in the design matrix and removed: "
            "the intercept is added by the model, not by the caller."
            % int(constant.sum()))
        X = X[:, ~constant]
    return X


class HipDesignMatrix():
    """Common part: handle ownership, counters, dot/Tdot through the C ABI."""

    use_cupy = False
    use_hip = True

    def __init__(self):
        self._h = c_void_p()
        self._lib = _lib.load()
        self.memoized = False
        self.X_dot_v = None
        self.v_prev = None
        self._count_offset = [0, 0]

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value and not getattr(_lib, "finalizing", True):
            try:
                self._lib.bbx_design_destroy(h)
            except Exception:
                pass
            self._h = c_void_p()

    @property
    def handle(self):
        return self._h

    @property
    def shape(self):
        n, P = c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_shape(self._h, byref(n), byref(P)))
        return int(n.value), int(P.value)

    @property
    def nnz(self):
        v = c_int64()
        _lib.check(self._lib.bbx_design_nnz(self._h, byref(v)))
        return int(v.value)

    @property
    def device(self):
        """HIP device index the operator lives on."""
        v = c_int()
        _lib.check(self._lib.bbx_design_device(self._h, byref(v)))
        return int(v.value)

    @property
    def storage_format(self):
        v = c_int()
        _lib.check(self._lib.bbx_design_format(self._h, byref(v)))
        return {0: "dense", 1: "csr", 2: "tiled"}[v.value]

    @property
    def storage_bytes(self):
        v = c_int64()
        _lib.check(self._lib.bbx_design_storage_bytes(self._h, byref(v)))
        return int(v.value)

    @property
    def matvec_bytes(self):
        """Algorithmic HBM bytes of (one dot, one Tdot)."""
        a, b = c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_matvec_bytes(self._h, byref(a),
                                                      byref(b)))
        return int(a.value), int(b.value)

    @property
    def timed_bytes(self):
        """Bytes moved by the kernels that get_timing() stamps (dot, Tdot)."""
        a, b = c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_timed_bytes(self._h, byref(a),
                                                     byref(b)))
        return int(a.value), int(b.value)

    @property
    def useful_bytes(self):
        """((dot, Tdot) bytes the timed kernels would move without padding and
        schedules, (padding share of the id stream of X, of X^T))."""
        from ctypes import c_double
        a, b = c_int64(), c_int64()
        pa, pb = c_double(), c_double()
        _lib.check(self._lib.bbx_design_useful_bytes(
            self._h, byref(a), byref(b), byref(pa), byref(pb)))
        return (int(a.value), int(b.value)), (pa.value, pb.value)

    @property
    def cg_launches(self):
        """Kernel launches per CG iteration on this design (3: direction step
        folded into the X~ v kernel, update into the Tdot epilogue)."""
        from ctypes import c_int
        v = c_int()
        _lib.check(self._lib.bbx_design_cg_launches(self._h, byref(v)))
        return int(v.value)

    def cg_stats(self, reset=False):
        """(CG solves, launches they enqueued past their stopping iteration,
        naps of the host between two stop tests) on this design since
        creation / the last reset."""
        from ctypes import c_int64
        a, b, c = c_int64(), c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_cg_stats(
            self._h, byref(a), byref(b), byref(c), 1 if reset else 0))
        return int(a.value), int(b.value), int(c.value)

    def set_cg_fold(self, on):
        """Direction step of the CG loop inside the X~ v kernel (3 launches per
        iteration) on / off for this design; None = the default (on for tiled
        value-free designs of up to 250 000 rows, where it measures faster;
        BBX_CG_FOLD=0|1 for the process)."""
        _lib.check(self._lib.bbx_design_set_cg_fold(
            self._h, -1 if on is None else int(bool(on))))

    @property
    def fused_operator_bytes(self):
        v = c_int64()
        _lib.check(self._lib.bbx_design_fused_operator_bytes(self._h,
                                                              byref(v)))
        return int(v.value)

    def tiled_info(self, chains=1):
        """Geometry of the LDS-tiled layout: {'X': {...}, 'Xt': {...}}; chains =
        2, 4: the layout sized for that many right-hand sides (once a batch of
        that width has been built).  'grid' = workgroups of one launch;
        'packed' = value-free ids stored as groups of five (csrc/
        tiled_layout.hpp) instead of four 16-bit ids per eight bytes."""
        out = {}
        base = {1: 0, 2: 2, 4: 4}[int(chains)]
        n, P = self.shape
        for which, name in ((base, 'X'), (base + 1, 'Xt')):
            W, nb, PR, G = c_int(), c_int(), c_int(), c_int()
            nq, ns = c_int64(), c_int64()
            pk = c_int()
            _lib.check(self._lib.bbx_design_tiled_info(
                self._h, which, byref(W), byref(nb), byref(PR), byref(G),
                byref(nq), byref(ns), byref(pk)))
            rows = n if name == 'X' else P - int(self.intercept_added)
            out[name] = dict(W=W.value, n_block=nb.value, PR=PR.value,
                             G=G.value, n_quad=nq.value, n_slice=ns.value,
                             packed=bool(pk.value), grid=-(-rows // max(PR.value, 1)) * G.value)
        return out

    @property
    def is_binary(self):
        """True for a sparse design whose stored entries all equal 1.0."""
        from ctypes import c_int
        f = c_int()
        _lib.check(self._lib.bbx_design_is_binary(self._h, byref(f)))
        return bool(f.value)

    @property
    def hybrid_info(self):
        """None, or how a mixed design was split by value in the tiled format:
        {'ones_nnz', 'rest_nnz', 'dense_nnz', 'dense_cols'} (entries equal to
        1.0 in the value-free layout, a dense block for columns full of other
        values, the rest in the valued layout)."""
        from ctypes import c_int
        hy, kd = c_int(), c_int()
        a, b, c = c_int64(), c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_hybrid_info(
            self._h, byref(hy), byref(a), byref(b), byref(c), byref(kd)))
        if not hy.value:
            return None
        return dict(ones_nnz=a.value, rest_nnz=b.value, dense_nnz=c.value,
                    dense_cols=kd.value)

    # -- the operator -------------------------------------------------------
    def dot(self, v):
        """X~ v (sparse_matrix.py:68-101, dense_matrix.py:37-48)."""
        if self.memoized:
            if np.all(self.v_prev == v):
                return self.X_dot_v
            self.v_prev = v.copy()
        v = np.ascontiguousarray(v, dtype=np.float64)
        n, P = self.shape
        if v.shape != (P,):
            raise ValueError("dot expects a vector of length %d" % P)
        out = np.empty(n, dtype=np.float64)
        _lib.check(self._lib.bbx_design_dot(self._h, _ptr(v), _ptr(out)))
        if self.memoized:
            self.X_dot_v = out
        return out

    def Tdot(self, w):
        """X~^T w (sparse_matrix.py:103-129, dense_matrix.py:50-52)."""
        w = np.ascontiguousarray(w, dtype=np.float64)
        n, P = self.shape
        if w.shape != (n,):
            raise ValueError("Tdot expects a vector of length %d" % n)
        out = np.empty(P, dtype=np.float64)
        _lib.check(self._lib.bbx_design_tdot(self._h, _ptr(w), _ptr(out)))
        return out

    def gram_matvec(self, obs_prec, v):
        """X~^T (obs_prec * (X~ v)) in one library call: the data part of the
        CG operator (cg_sampler.py:106-109), through the launches the CG loop
        uses."""
        n, P = self.shape
        w = np.ascontiguousarray(np.broadcast_to(
            np.asarray(obs_prec, dtype=np.float64), (n,)))
        v = np.ascontiguousarray(v, dtype=np.float64)
        if v.shape != (P,):
            raise ValueError("gram_matvec expects a vector of length %d" % P)
        out = np.empty(P, dtype=np.float64)
        _lib.check(self._lib.bbx_design_gram_matvec(
            self._h, _ptr(w), _ptr(v), _ptr(out)))
        return out

    def memoize_dot(self, flag=True):
        """abstract_matrix.py:41-47."""
        self.memoized = flag
        if self.v_prev is None:
            self.v_prev = np.full(self.shape[1], float('nan'))
        if not flag:
            self.X_dot_v = None
            self.v_prev = None

    # -- counters (abstract_matrix.py:61-72) --------------------------------
    def get_dot_count(self):
        a, b = c_int64(), c_int64()
        _lib.check(self._lib.bbx_design_matvec_count(self._h, byref(a),
                                                      byref(b)))
        return (int(a.value) + self._count_offset[0],
                int(b.value) + self._count_offset[1])

    @property
    def dot_count(self):
        return self.get_dot_count()[0]

    @property
    def Tdot_count(self):
        return self.get_dot_count()[1]

    @property
    def n_matvec(self):
        return sum(self.get_dot_count())

    def reset_matvec_count(self, count=0):
        if not hasattr(count, "__len__"):
            count = 2 * [count]
        _lib.check(self._lib.bbx_design_reset_matvec_count(self._h))
        self._count_offset = [int(count[0]), int(count[1])]

    # -- profiling ----------------------------------------------------------
    def set_timing(self, enabled=True, every=1):
        """HIP-event timing of the dot/Tdot launches; `every=N` samples one
        launch in N."""
        _lib.check(self._lib.bbx_design_set_timing(
            self._h, int(every) if enabled else 0))

    def reset_timing(self):
        _lib.check(self._lib.bbx_design_reset_timing(self._h))

    def get_timing(self):
        """{'dot': (launches, total_ms), 'tdot': (...)} from HIP events."""
        res = {}
        for which, name in ((0, "dot"), (1, "tdot"), (2, "operator")):
            cnt, ms = c_int64(), c_double()
            _lib.check(self._lib.bbx_design_get_timing(
                self._h, which, byref(cnt), byref(ms)))
            res[name] = (int(cnt.value), float(ms.value))
        return res

    def synchronize(self):
        _lib.check(self._lib.bbx_design_synchronize(self._h))

    def toarray(self):
        """The explicit n x P matrix X~ (abstract_matrix.py:66-68;
        dense_matrix.py:54-55).  Debugging aid: it is rebuilt column by column
        from P operator applications, the device never stores it."""
        n, P = self.shape
        out = np.empty((n, P))
        e = np.zeros(P)
        for j in range(P):
            e[j] = 1.
            out[:, j] = self.dot(e)
            e[j] = 0.
        return out

    def compute_fisher_info(self, weight, diag_only=False):
        raise NotImplementedError(
            "compute_fisher_info belongs to the 'cholesky' sampler, which is "
            "outside the CG hot path this backend implements.")

    def compute_transposed_fisher_info(self, weight, include_intrcpt=False):
        raise NotImplementedError(
            "outside the CG hot path this backend implements.")


class HipSparseDesignMatrix(HipDesignMatrix):
    """Counterpart of SparseDesignMatrix (sparse_matrix.py:18-49)."""

    def __init__(self, X, center_predictor=False, add_intercept=True,
                 copy_array=False, dot_format='csr', Tdot_format='csr',
                 device=0, storage='auto'):
        super().__init__()
        if dot_format == 'csc' or Tdot_format == 'csc':
            raise NotImplementedError(
                "Current dot operations are only implemented for the CSR "
                "format.")  # sparse_matrix.py:31-34
        if not sparse.issparse(X):
            raise TypeError("X must be a scipy sparse matrix")
        _lib.require_gpu()
        self.centered = center_predictor
        self.intercept_added = add_intercept
        X = remove_intercept_indicator(X)
        X = X.tocsr()
        if copy_array:
            X = X.copy()
        X.sort_indices()
        n, p = X.shape
        if center_predictor:
            self.column_offset = np.ascontiguousarray(
                np.squeeze(np.array(X.mean(axis=0))), dtype=np.float64
            ).reshape(p)
            offset = self.column_offset
        else:
            self.column_offset = np.zeros(p)
            offset = None
        data = np.ascontiguousarray(X.data, dtype=np.float64)
        self._create(n, p, X.indptr, X.indices, data, offset, add_intercept,
                     device, storage)

    def _create(self, n, p, indptr, indices, data, offset, add_intercept,
                device, storage):
        """int32 index arrays go to bbx_design_create_csr; 64-bit ones (what
        SciPy holds from 2^31 stored entries on) to bbx_design_create_csr64,
        which narrows them itself below that size."""
        fmt = {'auto': _lib.FORMAT_AUTO, 'csr': _lib.FORMAT_CSR,
               'tiled': _lib.FORMAT_TILED}[storage]
        # (either array 64-bit: both go over as int64 -- narrowing a row
        # pointer past 2^31 here would wrap silently)
        wide = (np.asarray(indptr).dtype.itemsize > 4
                or np.asarray(indices).dtype.itemsize > 4)
        dtype = np.int64 if wide else np.int32
        indptr = np.ascontiguousarray(indptr, dtype=dtype)
        indices = np.ascontiguousarray(indices, dtype=dtype)
        nnz = int(indptr[-1]) if len(indptr) else 0
        if len(indptr) != n + 1 or len(indices) < nnz:
            raise ValueError("indptr / indices do not describe %d rows" % n)
        if data is not None and len(data) < nnz:
            raise ValueError("data holds %d values, indptr[-1] says %d"
                             % (len(data), nnz))
        create = (self._lib.bbx_design_create_csr64 if wide
                  else self._lib.bbx_design_create_csr)
        _lib.check(create(
            n, p, nnz, _ptr(indptr), _ptr(indices), _ptr(data),
            _ptr(offset), int(bool(add_intercept)), int(device), fmt,
            byref(self._h)))

    @classmethod
    def from_csr_arrays(cls, shape, indptr, indices, data=None,
                        column_offset=None, add_intercept=True, device=0,
                        storage='auto'):
        """The design from raw CSR arrays in host memory, int32 or int64 --
        for matrices too large to also hold SciPy's float64 `data` of a
        binary design (`data=None`: every stored value is 1.0).  Column ids
        must ascend within a row; `column_offset` (p) centres the columns.
        Unlike the main constructor (abstract_matrix.py:93-107) this path does
        NOT look for constant columns: a column of ones next to
        `add_intercept=True` makes the design rank-deficient -- drop it
        before calling."""
        self = cls.__new__(cls)
        HipDesignMatrix.__init__(self)
        _lib.require_gpu()
        n, p = int(shape[0]), int(shape[1])
        self.centered = column_offset is not None
        self.intercept_added = add_intercept
        if column_offset is None:
            self.column_offset = np.zeros(p)
            offset = None
        else:
            offset = self.column_offset = np.ascontiguousarray(
                column_offset, dtype=np.float64).reshape(p)
        if data is not None:
            data = np.ascontiguousarray(data, dtype=np.float64)
        self._create(n, p, indptr, indices, data, offset, add_intercept,
                     device, storage)
        return self

    @classmethod
    def from_device_csr(cls, n, p, nnz, indptr_ptr, indices_ptr, data_ptr=None,
                        offset_ptr=None, add_intercept=True, device=0,
                        storage='auto'):
        """Adopts CSR arrays that already live in HBM (raw device pointers,
        e.g. torch tensors' data_ptr()); the arrays are copied."""
        self = cls.__new__(cls)
        HipDesignMatrix.__init__(self)
        _lib.require_gpu()
        self.centered = offset_ptr is not None
        self.intercept_added = add_intercept
        self.column_offset = None
        fmt = {'auto': _lib.FORMAT_AUTO, 'csr': _lib.FORMAT_CSR,
               'tiled': _lib.FORMAT_TILED}[storage]
        _lib.check(self._lib.bbx_design_create_csr_dev(
            n, p, nnz, c_void_p(indptr_ptr), c_void_p(indices_ptr),
            c_void_p(data_ptr) if data_ptr else None,
            c_void_p(offset_ptr) if offset_ptr else None,
            int(bool(add_intercept)), int(device), fmt, byref(self._h)))
        return self

    @property
    def is_sparse(self):
        return True


class HipDenseDesignMatrix(HipDesignMatrix):
    """Counterpart of DenseDesignMatrix (dense_matrix.py:7-27).  Unlike the
    reference, the caller's array is never modified (the reference centres it
    in place unless copy_array=True, dense_matrix.py:17-22)."""

    def __init__(self, X, center_predictor=False, add_intercept=True,
                 copy_array=False, device=0, storage_dtype='float64'):
        super().__init__()
        _lib.require_gpu()
        X = np.asarray(X)
        X = remove_intercept_indicator(X)
        if X.dtype == np.float32:
            in_dtype = _lib.F32
        else:
            X = np.asarray(X, dtype=np.float64)
            in_dtype = _lib.F64
        X = np.ascontiguousarray(X)
        n, p = X.shape
        self.centered = center_predictor
        self.intercept_added = add_intercept
        offset = None
        if center_predictor:
            offset = np.ascontiguousarray(
                np.mean(X, axis=0, dtype=np.float64), dtype=np.float64)
        self.column_offset = offset
        st = {'float64': _lib.F64, 'float32': _lib.F32}[storage_dtype]
        self.storage_dtype = storage_dtype
        _lib.check(self._lib.bbx_design_create_dense(
            n, p, _ptr(X), in_dtype, st, _ptr(offset),
            int(bool(add_intercept)), int(device), byref(self._h)))

    @classmethod
    def from_device_array(cls, n, p, X_ptr, offset_ptr=None,
                          add_intercept=True, device=0, in_dtype='float32',
                          storage_dtype='float32'):
        """Adopts a row-major n x p array that already lives in HBM (raw
        device pointer); it is copied (centred, intercept column added) into
        the operator's own storage."""
        self = cls.__new__(cls)
        HipDesignMatrix.__init__(self)
        _lib.require_gpu()
        self.centered = offset_ptr is not None
        self.intercept_added = add_intercept
        self.column_offset = None
        self.storage_dtype = storage_dtype
        code = {'float64': _lib.F64, 'float32': _lib.F32}
        _lib.check(self._lib.bbx_design_create_dense_dev(
            n, p, c_void_p(X_ptr), code[in_dtype], code[storage_dtype],
            c_void_p(offset_ptr) if offset_ptr else None,
            int(bool(add_intercept)), int(device), byref(self._h)))
        return self

    @property
    def is_sparse(self):
        return False

    @property
    def nnz(self):
        n, P = self.shape
        return n * P
