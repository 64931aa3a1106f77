"""Bridge prior on the regression coefficients.

Scalar set-up only (no data-parallel work).  The constructor keywords, the
attribute names the Gibbs driver reads and the keys of `get_info()` are those
of the reference's `RegressionCoefPrior` (prior.py:7-208) so that user scripts
carry over; the implementation is this package's own.

Conventions (prior.py:129-167): the sampler runs in the *raw* parametrisation
pi(beta_j | tau, lambda_j) prop. to exp(-|beta_j / (tau lambda_j)|^alpha) summed
out to a scale tau; users see tau multiplied by E|beta| of the unit-scale
exponential-power density, m(alpha) = Gamma(2/alpha) / Gamma(1/alpha)
('coef_magnitude').
"""
import math
from warnings import warn

import numpy as np
from scipy.optimize import brentq
from scipy.special import digamma, polygamma

_INFO_KEYS = (
    'bridge_exponent', 'n_fixed_effect', 'sd_for_intercept',
    'sd_for_fixed_effect', 'regularizing_slab_size',
    'global_scale_prior_hyper_param', '_global_scale_parametrization')
_LN10 = math.log(10.)


def unit_magnitude(alpha):
    """m(alpha) = E|x| under density prop. to exp(-|x|^alpha)."""
    return math.gamma(2. / alpha) / math.gamma(1. / alpha)


def gamma_hyperparameters(log10_mean, log10_sd, alpha, parametrization):
    """Gamma(shape, rate) on phi = tau^-alpha such that log10(tau) has the
    requested mean and sd (what prior.py:143-208 solves).

    With phi ~ Gamma(a, b): log(tau) = -log(phi)/alpha, so
        sd(log tau)   = sqrt(trigamma(a)) / alpha      -> fixes a
        mean(log tau) = (log b - digamma(a)) / alpha   -> fixes b given a.
    trigamma decreases monotonically from +inf to 0, so the first equation has
    one root; it is bracketed on a log grid and polished by Brent's method.
    """
    if log10_sd < 0:
        raise ValueError("Variance has to be positive.")
    if log10_sd > 1e8 / _LN10:
        raise ValueError("Specified prior variance is too large.")
    target_var = (alpha * log10_sd * _LN10) ** 2
    mean_log_tau = log10_mean * _LN10
    if parametrization == 'coef_magnitude':
        mean_log_tau -= math.log(unit_magnitude(alpha))

    def gap(u):  # u = log(shape)
        return float(polygamma(1, math.exp(u))) - target_var

    lo, hi = -10., 10.
    if gap(lo) < 0:
        raise ValueError("Requested prior sd of log10(global scale) is too "
                         "large to be matched by a Gamma prior.")
    while gap(hi) > 0:
        lo, hi = hi, hi + 10.
        if hi > 1e4:
            raise ValueError("Requested prior sd of log10(global scale) is "
                             "too small to be matched by a Gamma prior.")
    shape = math.exp(brentq(gap, lo, hi, xtol=1e-13, rtol=1e-13))
    rate = math.exp(float(digamma(shape)) + alpha * mean_log_tau)
    return shape, rate


class RegressionCoefPrior():

    def __init__(self, bridge_exponent=.5, n_fixed_effect=0,
                 sd_for_intercept=float('inf'),
                 sd_for_fixed_effect=float('inf'),
                 regularizing_slab_size=float('inf'),
                 global_scale_prior_hyper_param=None,
                 _global_scale_parametrization='coef_magnitude'):
        if bridge_exponent > 2:
            raise ValueError("Exponent larger than 2 is unsupported.")
        sd_fixed = np.atleast_1d(
            np.asarray(sd_for_fixed_effect, dtype=np.float64))
        if np.ndim(sd_for_fixed_effect) == 0:
            sd_fixed = np.full(n_fixed_effect, float(sd_for_fixed_effect))
        elif sd_fixed.size != n_fixed_effect:
            raise ValueError(
                "Prior sd for fixed effects must be specified either by a "
                "scalar or array of the same length as n_fixed_effect.")
        self.bridge_exp = bridge_exponent
        self.n_fixed = n_fixed_effect
        self.sd_for_intercept = sd_for_intercept
        self.sd_for_fixed = sd_fixed
        self.slab_size = regularizing_slab_size
        self._gscale_paramet = _global_scale_parametrization
        # param['gscale_neg_power']: Gamma(shape, rate) on tau^-alpha read by
        # the tau update; (0, 0) is the scale-invariant reference prior.
        hyper = global_scale_prior_hyper_param
        shape, rate, moments = 0., 0., None
        if hyper is not None:
            try:
                moments = {k: hyper[k] for k in ('log10_mean', 'log10_sd')}
            except KeyError:
                raise ValueError("Dictionary should contain keys 'log10_mean' "
                                 "and 'log10_sd.'") from None
            shape, rate = gamma_hyperparameters(
                moments['log10_mean'], moments['log10_sd'], bridge_exponent,
                _global_scale_parametrization)
        self.param = {'gscale_neg_power': {'shape': shape, 'rate': rate},
                      'gscale': moments}

    # --- introspection / copying -------------------------------------------
    def get_info(self):
        sd = self.sd_for_fixed
        uniform = sd.size > 0 and bool(np.all(sd == sd[0]))
        values = (self.bridge_exp, self.n_fixed, self.sd_for_intercept,
                  sd[0] if uniform else sd, self.slab_size,
                  self.param['gscale'], self._gscale_paramet)
        return dict(zip(_INFO_KEYS, values))

    def clone(self, **changes):
        if '_global_scale_parametrization' in changes:
            raise ValueError("Change of parametrization is not supported.")
        spec = self.get_info()
        unknown = [k for k in changes if k not in spec]
        for key in unknown:
            warn("'%s' is not a keyword of RegressionCoefPrior; ignored." % key)
        spec.update({k: v for k, v in changes.items() if k in spec})
        return RegressionCoefPrior(**spec)

    # --- parametrisation ------------------------------------------------------
    @staticmethod
    def compute_power_exp_ave_magnitude(exponent, scale=1.):
        return scale * unit_magnitude(exponent)

    def adjust_scale(self, gscale, lscale, to):
        """Moves (tau, lambda) between 'raw' and 'coef_magnitude'; tau*lambda is
        invariant.  Array arguments are updated in place (callers rely on it
        for the stored samples, bayesbridge.py:244-251)."""
        m = unit_magnitude(self.bridge_exp)
        if to not in ('raw', 'coef_magnitude'):
            raise ValueError("to must be 'raw' or 'coef_magnitude'")
        # (numerator, denominator) applied to tau; lambda gets the inverse
        up, down = (m, 1.) if to == 'coef_magnitude' else (1., m)

        def rescale(value, mul, div):
            if isinstance(value, np.ndarray):
                if mul != 1.:
                    value *= mul
                if div != 1.:
                    value /= div
                return value
            return value * mul / div if mul != 1. else value / div

        return rescale(gscale, up, down), rescale(lscale, down, up)

    def solve_for_gscale_prior_hyperparam(self, log10_mean, log10_sd,
                                          bridge_exp, gscale_paramet):
        return gamma_hyperparameters(log10_mean, log10_sd, bridge_exp,
                                     gscale_paramet)
