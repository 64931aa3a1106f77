"""Prior specification; same constructor and attributes as the reference's
`RegressionCoefPrior` (prior.py:7-111) so user scripts carry over.  Pure
scalar set-up, no data-parallel work."""
import math
from warnings import warn

import numpy as np
from scipy.optimize import brentq
from scipy.special import polygamma


class RegressionCoefPrior():

    def __init__(
            self,
            bridge_exponent=.5,
            n_fixed_effect=0,
            sd_for_intercept=float('inf'),
            sd_for_fixed_effect=float('inf'),
            regularizing_slab_size=float('inf'),
            global_scale_prior_hyper_param=None,
            _global_scale_parametrization='coef_magnitude'):
        if not (np.isscalar(sd_for_fixed_effect)
                or n_fixed_effect == len(sd_for_fixed_effect)):
            raise ValueError(
                "Prior sd for fixed effects must be specified either by a "
                "scalar or array of the same length as n_fixed_effect.")
        if bridge_exponent > 2:
            raise ValueError("Exponent larger than 2 is unsupported.")
        if np.isscalar(sd_for_fixed_effect):
            sd_for_fixed_effect = sd_for_fixed_effect * np.ones(n_fixed_effect)
        self.sd_for_intercept = sd_for_intercept
        self.sd_for_fixed = np.asarray(sd_for_fixed_effect, dtype=np.float64)
        self.slab_size = regularizing_slab_size
        self.n_fixed = n_fixed_effect
        self.bridge_exp = bridge_exponent
        self._gscale_paramet = _global_scale_parametrization
        if global_scale_prior_hyper_param is None:
            # reference prior of a scale family (prior.py:77-81)
            self.param = {'gscale_neg_power': {'shape': 0., 'rate': 0.},
                          'gscale': None}
        else:
            keys = global_scale_prior_hyper_param.keys()
            if not ({'log10_mean', 'log10_sd'} <= keys):
                raise ValueError(
                    "Dictionary should contain keys 'log10_mean' and 'log10_sd.'")
            log10_mean = global_scale_prior_hyper_param['log10_mean']
            log10_sd = global_scale_prior_hyper_param['log10_sd']
            shape, rate = self.solve_for_gscale_prior_hyperparam(
                log10_mean, log10_sd, bridge_exponent, self._gscale_paramet)
            self.param = {
                'gscale_neg_power': {'shape': shape, 'rate': rate},
                'gscale': {'log10_mean': log10_mean, 'log10_sd': log10_sd}}

    def get_info(self):
        sd_for_fixed = self.sd_for_fixed
        if len(sd_for_fixed) > 0 and np.all(sd_for_fixed == sd_for_fixed[0]):
            sd_for_fixed = sd_for_fixed[0]
        return {
            'bridge_exponent': self.bridge_exp,
            'n_fixed_effect': self.n_fixed,
            'sd_for_intercept': self.sd_for_intercept,
            'sd_for_fixed_effect': sd_for_fixed,
            'regularizing_slab_size': self.slab_size,
            'global_scale_prior_hyper_param': self.param['gscale'],
            '_global_scale_parametrization': self._gscale_paramet,
        }

    def clone(self, **kwargs):
        info = self.get_info()
        if '_global_scale_parametrization' in kwargs:
            raise ValueError("Change of parametrization is not supported.")
        for key, value in kwargs.items():
            if key in info:
                info[key] = value
            else:
                warn("'{:s} is not a valid keyward argument.".format(key))
        return RegressionCoefPrior(**info)

    def adjust_scale(self, gscale, lscale, to):
        """prior.py:129-141.  In-place on array arguments, like the reference."""
        unit = self.compute_power_exp_ave_magnitude(self.bridge_exp, 1.)
        if to == 'raw':
            gscale /= unit
            lscale *= unit
        elif to == 'coef_magnitude':
            gscale *= unit
            lscale /= unit
        else:
            raise ValueError()
        return gscale, lscale

    @staticmethod
    def compute_power_exp_ave_magnitude(exponent, scale=1.):
        """E|x| for density prop. to exp(-|x/scale|^exponent) (prior.py:163-167)."""
        return scale * math.gamma(2 / exponent) / math.gamma(1 / exponent)

    def solve_for_gscale_prior_hyperparam(
            self, log10_mean, log10_sd, bridge_exp, gscale_paramet):
        """Gamma(shape, rate) on tau^-bridge_exp matching the requested mean and
        sd of log10(tau) (prior.py:143-208)."""
        log_mean = log10_mean * math.log(10.)
        log_sd = log10_sd * math.log(10.)
        if gscale_paramet == 'coef_magnitude':
            log_mean -= math.log(
                self.compute_power_exp_ave_magnitude(bridge_exp, 1.))
        if log_sd < 0:
            raise ValueError("Variance has to be positive.")
        if log_sd > 10 ** 8:
            raise ValueError("Specified prior variance is too large.")

        def excess_sd(log_shape):
            trigamma = float(polygamma(1, math.exp(log_shape)))
            return math.sqrt(trigamma) / bridge_exp - log_sd

        lower = -10.
        if excess_sd(lower) < 0:
            raise ValueError("Objective function must have positive value "
                             "at the lower limit.")
        while excess_sd(lower + 5.) > 0 and lower < 10 ** 4:
            lower += 5.
        log_shape = brentq(excess_sd, lower, lower + 5.)
        shape = math.exp(log_shape)
        rate = math.exp(float(polygamma(0, shape)) + bridge_exp * log_mean)
        return shape, rate
