"""Oracle restatement of the design-matrix operator (test infrastructure).

Follows design_matrix/sparse_matrix.py:21-129 and dense_matrix.py:9-52 of the
reference: the represented matrix is  X~ = [1 | X - 1 offset^T]  with
    dot(v)  = v0 + X v1 - <offset, v1>            (sparse_matrix.py:77-81,90-98)
    Tdot(w) = [sum w ; X^T w - sum(w) offset]     (sparse_matrix.py:108-114,121-129)
The dense class applies centring and the ones column to the array itself
(dense_matrix.py:21-25) and then uses plain products (dense_matrix.py:42,52).
"""
import numpy as np
import scipy.sparse as sp


def drop_constant_columns(X):
    """abstract_matrix.py:93-107: columns with variance < n * 2^-52 go."""
    if sp.issparse(X):
        m1 = np.asarray(X.mean(axis=0)).ravel()
        m2 = np.asarray(X.power(2).mean(axis=0)).ravel()
        var = m2 - m1 ** 2
    else:
        var = np.var(X, axis=0)
    keep = ~(var < X.shape[0] * 2.0 ** -52)
    if not keep.all():
        X = X[:, keep]
    return X


class _Counting:
    use_cupy = False

    def __init__(self):
        self.dot_count = 0
        self.Tdot_count = 0

    def get_dot_count(self):
        return self.dot_count, self.Tdot_count

    @property
    def n_matvec(self):
        return self.dot_count + self.Tdot_count


class OracleSparseDesign(_Counting):

    def __init__(self, X, center_predictor=False, add_intercept=True):
        super().__init__()
        X = drop_constant_columns(sp.csr_matrix(X))
        self.X_main = X.tocsr()
        self.X_main_T = self.X_main.T  # CSC view, as `X.T.dot` in the reference
        self.centered = center_predictor
        self.intercept_added = add_intercept
        p = X.shape[1]
        if center_predictor:
            self.column_offset = np.asarray(X.mean(axis=0)).ravel()
        else:
            self.column_offset = np.zeros(p)

    @property
    def shape(self):
        n, p = self.X_main.shape
        return n, p + int(self.intercept_added)

    @property
    def is_sparse(self):
        return True

    @property
    def nnz(self):
        return self.X_main.nnz

    def dot(self, v):
        v = np.asarray(v, dtype=np.float64)
        lead = 0.0
        if self.intercept_added:
            lead = v[0]
            v = v[1:]
        out = self.X_main.dot(v)
        out -= np.inner(self.column_offset, v)
        self.dot_count += 1
        return lead + out

    def Tdot(self, w):
        w = np.asarray(w, dtype=np.float64)
        total = np.sum(w)
        g = self.X_main_T.dot(w)
        g -= total * self.column_offset
        if self.intercept_added:
            g = np.concatenate(([total], g))
        self.Tdot_count += 1
        return g

    def toarray(self):
        """Explicit X~ (for tiny test cases only)."""
        A = self.X_main.toarray() - self.column_offset[None, :]
        if self.intercept_added:
            A = np.hstack((np.ones((A.shape[0], 1)), A))
        return A


class OracleDenseDesign(_Counting):

    def __init__(self, X, center_predictor=False, add_intercept=True):
        super().__init__()
        X = np.array(X, dtype=np.float64, copy=True)
        X = drop_constant_columns(X)
        if center_predictor:
            X = X - np.mean(X, axis=0)[None, :]
        if add_intercept:
            X = np.hstack((np.ones((X.shape[0], 1)), X))
        self.X = X
        self.centered = center_predictor
        self.intercept_added = add_intercept

    @classmethod
    def from_full(cls, X_full, centered=True):
        """Wraps an [n, P] float64 array that already IS the operator: column
        0 the intercept's ones, the others centred (what
        DenseDesignMatrix.__init__ leaves behind, dense_matrix.py:9-27) --
        without the copies the constructor makes (bench.py's config-4 leg:
        12.8 GB)."""
        self = cls.__new__(cls)
        _Counting.__init__(self)
        assert X_full.dtype == np.float64 and X_full.ndim == 2
        self.X = X_full
        self.centered = centered
        self.intercept_added = True
        return self

    @property
    def shape(self):
        return self.X.shape

    @property
    def is_sparse(self):
        return False

    @property
    def nnz(self):
        return self.X.size

    def dot(self, v):
        self.dot_count += 1
        return self.X.dot(v)

    def Tdot(self, w):
        self.Tdot_count += 1
        return self.X.T.dot(w)

    def toarray(self):
        return self.X


def make_design(X, add_intercept=True, center_predictor=True):
    """model/factory.py:10-52 for linear/logit: sparse input -> sparse operator,
    defaults add_intercept=True, center_predictor=True."""
    if isinstance(X, _Counting):      # an operator built by the caller
        return X
    cls = OracleSparseDesign if sp.issparse(X) else OracleDenseDesign
    return cls(X, center_predictor=center_predictor,
               add_intercept=add_intercept)
