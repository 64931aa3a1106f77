"""Likelihood models on top of the HIP design operator: the subset of the
reference's model layer that the 'cg' Gibbs path touches
(model/factory.py:10-68, linear_model.py:6-45, logistic_model.py:6-116).
The Cox model only supports HMC in the reference (gibbs_util.py:76-80) and is
outside this backend."""
import math
from warnings import warn

import numpy as np
import scipy.sparse as sparse

from .design_matrix import (HipDenseDesignMatrix, HipDesignMatrix,
                            HipSparseDesignMatrix)


class _Model():

    @property
    def n_obs(self):
        return self.design.shape[0]

    @property
    def n_pred(self):
        return self.design.shape[1]

    @property
    def intercept_added(self):
        return self.design.intercept_added


class LinearModel(_Model):

    def __init__(self, y, design):
        self.y = np.asarray(y, dtype=np.float64)
        self.design = design
        self.name = 'linear'
        if len(self.y) != design.shape[0]:
            raise ValueError(
                "Incompatible sizes of the outcome and design matrix.")

    def compute_loglik_and_gradient(self, beta, obs_prec, loglik_only=False):
        X_beta = self.design.dot(beta)              # linear_model.py:13-22
        loglik = (len(self.y) * math.log(obs_prec) / 2
                  - obs_prec * np.sum((self.y - X_beta) ** 2) / 2)
        grad = None
        if not loglik_only:
            grad = obs_prec * self.design.Tdot(self.y - X_beta)
        return loglik, grad

    def calc_intercept_mle(self):
        return self.y.mean()

    @staticmethod
    def simulate_outcome(X, beta, noise_sd, seed=None):
        if seed is not None:
            np.random.seed(seed)
        return X.dot(beta) + noise_sd * np.random.randn(X.shape[0])


class LogisticModel(_Model):

    def __init__(self, n_success, n_trial, design):
        """Outcome checks of logistic_model.py:10-47: counts must line up with
        the rows of the design, 0 < n_trial, n_success <= n_trial; without
        n_trial the outcome has to be 0/1."""
        n_row = design.shape[0]
        n_success = np.asarray(n_success, dtype=np.float64)
        binary_default = n_trial is None
        if binary_default:
            if n_success.size and n_success.max() > 1:
                raise ValueError(
                    "n_trial is required when the outcome is not binary.")
            n_trial = np.ones(n_success.shape)
        else:
            n_trial = np.asarray(n_trial, dtype=np.float64)
        if not (n_success.shape == n_trial.shape == (n_row,)):
            raise ValueError(
                "Outcome vectors and design matrix have incompatible sizes: "
                "%s successes, %s trials, %d rows."
                % (n_success.shape, n_trial.shape, n_row))
        if not binary_default:
            if (n_trial <= 0).any():
                raise ValueError("Every n_trial must be strictly positive.")
            if (n_success > n_trial).any():
                raise ValueError("n_success exceeds n_trial in some row.")
        else:
            warn("n_trial not given: treating the outcome as binary.")
        self.n_success = n_success
        self.n_trial = n_trial
        self.design = design
        self.name = 'logit'

    def compute_loglik_and_gradient(self, beta, loglik_only=False):
        logit_prob = self.design.dot(beta)           # logistic_model.py:49-60
        loglik = np.sum(self.n_success * logit_prob
                        - self.n_trial * np.logaddexp(0, logit_prob))
        grad = None
        if not loglik_only:
            prob = 1 / (1 + np.exp(-logit_prob))
            grad = self.design.Tdot(self.n_success - self.n_trial * prob)
        return loglik, grad

    def calc_intercept_mle(self):
        p_hat = self.n_success.mean() / self.n_trial.mean()
        return np.log(p_hat / (1 - p_hat))

    @staticmethod
    def compute_polya_gamma_mean(shape, tilt):
        """logistic_model.py:80-87 (including its b/2 value at |tilt| <= 1e-5)."""
        pg_mean = shape.copy() / 2
        nz = (np.abs(tilt) > 1e-5)
        pg_mean[nz] *= 1 / tilt[nz] * (np.exp(tilt[nz]) - 1) \
            / (np.exp(tilt[nz]) + 1)
        return pg_mean

    @staticmethod
    def simulate_outcome(n_trial, X, beta, seed=None):
        prob = 1 / (1 + np.exp(-X.dot(beta)))
        if seed is not None:
            np.random.seed(seed)
        return np.random.binomial(n_trial, prob)


def RegressionModel(outcome, X, family='linear', add_intercept=None,
                    center_predictor=True, device=0, storage='auto',
                    dense_storage_dtype='float64'):
    """model/factory.py:10-68 with the design placed on an MI355X.  `X` may be
    a SciPy sparse matrix, a NumPy array, or an already built HipDesignMatrix."""
    if family == 'cox':
        raise NotImplementedError(
            "The Cox model uses the HMC sampler, which is outside the CG hot "
            "path this backend implements.")
    if add_intercept is None:
        add_intercept = True
    if isinstance(X, HipDesignMatrix):
        design = X
    elif sparse.issparse(X):
        design = HipSparseDesignMatrix(
            X, add_intercept=add_intercept, center_predictor=center_predictor,
            device=device, storage=storage)
    else:
        design = HipDenseDesignMatrix(
            X, add_intercept=add_intercept, center_predictor=center_predictor,
            device=device, storage_dtype=dense_storage_dtype)
    if family == 'linear':
        return LinearModel(outcome, design)
    if family == 'logit':
        if isinstance(outcome, tuple):
            n_success, n_trial = outcome
        else:
            n_success, n_trial = outcome, None
        return LogisticModel(n_success, n_trial, design)
    raise NotImplementedError()
