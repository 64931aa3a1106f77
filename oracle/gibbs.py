"""Oracle restatement of one whole Gibbs chain on the 'cg' path (test
infrastructure; also the `cpu_baseline` of bench.py).

Follows BayesBridge.gibbs (bayesbridge.py:109-277) for the linear and logit
families with coef_sampler_type='cg': seeding (random/random.py:17-22),
initialisation without mode search when `init` holds 'coef'
(bayesbridge.py:279-353), and per iteration (bayesbridge.py:210-240)
    beta | rest   reg_coef_sampler.py:60-103 -> oracle.cg_sample
    Omega | beta  bayesbridge.py:397-410
    tau | beta    bayesbridge.py:412-448
    lambda | ...  bayesbridge.py:458-478
    logp          bayesbridge.py:480-511
The initial mode search of the reference (reg_coef_sampler.py:281-391, SciPy
L-BFGS-B) is restated in `search_mode`.  `record=True` keeps the inputs and
outputs of every CG draw, which is how the per-iteration golden fixtures are
made and replayed.
"""
import math

import numpy as np
import scipy.optimize

from .cg_sampler import cg_sample
from .design_matrix import make_design
from .rng import OracleRandom
from .summarizer import CoefSummarizer, regularized_prior_scale


def unit_bridge_magnitude(exponent):
    """prior.py:163-167 with scale 1."""
    return math.gamma(2 / exponent) / math.gamma(1 / exponent)


def pg_mean(shape, tilt):
    """logistic_model.py:80-87."""
    out = shape.copy() / 2
    nz = np.abs(tilt) > 1e-5
    out[nz] *= 1 / tilt[nz] * (np.exp(tilt[nz]) - 1) / (np.exp(tilt[nz]) + 1)
    return out


def loglik(family, design, outcome, coef, obs_prec):
    eta = design.dot(coef)
    if family == 'linear':                               # linear_model.py:13-17
        y = outcome
        return len(y) * math.log(obs_prec) / 2 \
            - obs_prec * np.sum((y - eta) ** 2) / 2
    n_success, n_trial = outcome                         # logistic_model.py:49-55
    return np.sum(n_success * eta - n_trial * np.logaddexp(0, eta))


def loglik_grad(family, design, outcome, coef, obs_prec):
    eta = design.dot(coef)
    if family == 'linear':
        return obs_prec * design.Tdot(outcome - eta)
    n_success, n_trial = outcome
    prob = 1 / (1 + np.exp(-eta))
    return design.Tdot(n_success - n_trial * prob)


def search_mode(family, design, outcome, coef, lscale, gscale, obs_prec,
                sd_unshrunk, slab, maxiter=250):
    """reg_coef_sampler.py:281-391, default (L-BFGS-B) branch."""
    P, nu = len(coef), len(sd_unshrunk)
    scale = np.ones(P)
    scale[nu:] = regularized_prior_scale(gscale, lscale, slab)
    with np.errstate(divide='ignore'):
        pprec = np.concatenate(((sd_unshrunk / scale[:nu]) ** -2,
                                np.ones(P - nu)))

    def fun(theta):
        return -(loglik(family, design, outcome, theta * scale, obs_prec)
                 + np.sum(-pprec * theta ** 2) / 2)

    def jac(theta):
        g = loglik_grad(family, design, outcome, theta * scale, obs_prec)
        return -(scale * g - pprec * theta)

    res = scipy.optimize.minimize(
        fun, coef / scale, method='L-BFGS-B', jac=jac,
        options={'maxiter': maxiter, 'gtol': 10 ** -6 / np.sqrt(P),
                 'maxcor': 200})
    return scale * res.x, res


class OracleGibbs:

    def __init__(self, outcome, X, family, bridge_exponent=.5,
                 sd_for_intercept=float('inf'),
                 regularizing_slab_size=float('inf'),
                 gscale_shape=0., gscale_rate=0., add_intercept=True,
                 center_predictor=True, use_scipy_cg=False, omp_threads=None):
        self.family = family
        if omp_threads is not None:
            # multi-core baseline: OpenMP products and CG (omp_baseline.py)
            from .omp_baseline import OmpSparseDesign
            self.design = OmpSparseDesign(
                X, center_predictor=center_predictor,
                add_intercept=add_intercept, n_threads=omp_threads)
        else:
            self.design = make_design(X, add_intercept=add_intercept,
                                      center_predictor=center_predictor)
        if family == 'logit':
            n_success, n_trial = outcome
            self.outcome = (np.asarray(n_success, dtype=np.float64),
                            np.asarray(n_trial, dtype=np.float64))
        else:
            self.outcome = np.asarray(outcome, dtype=np.float64)
        self.n, self.P = self.design.shape
        self.alpha = bridge_exponent
        self.slab = regularizing_slab_size
        self.sd_unshrunk = np.array([sd_for_intercept]) if add_intercept \
            else np.zeros(0)
        self.nu = len(self.sd_unshrunk)
        self.shape0, self.rate0 = gscale_shape, gscale_rate
        self.use_scipy_cg = use_scipy_cg
        self.rng = None

    # --- conditional updates -------------------------------------------------
    def draw_obs_prec(self, coef):
        eta = self.design.dot(coef)
        if self.family == 'linear':
            scale = np.sum((self.outcome - eta) ** 2) / 2
            return 1 / (scale / np.random.gamma(self.n / 2, 1))
        return self.rng.polya_gamma(self.outcome[1].astype(np.intc), eta)

    def draw_gscale(self, beta):
        if beta.size == 0:
            return 1.
        if np.count_nonzero(beta) == 0:
            g = 0
        else:
            shape = self.shape0 + beta.size / self.alpha
            rate = self.rate0 + np.sum(np.abs(beta) ** self.alpha)
            phi = np.random.gamma(shape, scale=1 / rate)
            g = 1 / phi ** (1 / self.alpha)
        return max(g, .001 / unit_bridge_magnitude(self.alpha))

    def draw_lscale(self, gscale, beta):
        if self.alpha == 2:
            return .5 * np.ones(beta.size)
        ls = np.sqrt(.5 / self.rng.tilted_stable(
            self.alpha / 2, (beta / gscale) ** 2))
        if np.any(ls == 0):
            ls[ls == 0] = 10e-16
        elif np.any(np.isinf(ls)):
            ls[np.isinf(ls)] = 2.0 / gscale
        return ls

    def logp(self, coef, gscale, obs_prec):
        nu = self.nu
        lp = loglik(self.family, self.design, self.outcome, coef, obs_prec)
        lp += - .5 * np.sum((coef / self.slab) ** 2)
        lp += - (len(coef) - nu) * math.log(gscale) \
            - np.sum(np.abs(coef[nu:] / gscale) ** self.alpha)
        sd = self.sd_unshrunk
        lp += - 1 / 2 * np.sum((coef[:nu] / sd) ** 2)
        lp += - np.sum(np.log(sd[sd < float('inf')]))
        lp += (self.shape0 - 1.) * math.log(gscale) - self.rate0 * gscale
        return lp

    def draw_coef(self, obs_prec, gscale, lscale, summ, record=None):
        if self.family == 'linear':                      # bayesbridge.py:376-380
            y_gauss = self.outcome
            omega = obs_prec * np.ones(self.n)
        else:
            omega = obs_prec
            y_gauss = (self.outcome[0] - self.outcome[1] / 2) / obs_prec
        z = self.design.Tdot(omega * y_gauss)            # reg_coef_sampler.py:74
        prior_sd = np.concatenate((
            self.sd_unshrunk,
            regularized_prior_scale(gscale, lscale, self.slab)))
        with np.errstate(divide='ignore'):
            phi = 1 / prior_sd
        x0 = summ.extrapolate_coef_condmean(gscale, lscale)
        sd = summ.estimate_post_sd()
        eta1 = np.random.randn(self.n)                   # cg_sampler.py:61-62
        eta2 = np.random.randn(self.P)
        atol = 10e-6 * np.sqrt(self.P)
        if hasattr(self.design, 'cg_sample'):
            coef, info = self.design.cg_sample(omega, phi, z, x0, sd, self.nu,
                                               eta1, eta2, 500, atol)
        else:
            coef, info = cg_sample(self.design, omega, phi, z, x0, sd, self.nu,
                                   eta1, eta2, 500, atol,
                                   use_scipy=self.use_scipy_cg)
        summ.update(coef, gscale, lscale)
        if record is not None:
            record.append(dict(obs_prec=omega.copy(), prior_prec_sqrt=phi,
                               z=z, coef_cg_init=x0, coef_scaled_sd=sd.copy(),
                               randn_n=eta1, randn_P=eta2, coef=coef.copy(),
                               n_iter=info['n_iter'], gscale=gscale,
                               lscale=lscale.copy()))
        return coef, info

    # --- driver --------------------------------------------------------------
    def gibbs(self, n_iter, seed=None, init=None, record=False,
              gscale_parametrization='coef_magnitude'):
        init = dict(init or {'global_scale': .1})
        self.rng = OracleRandom(seed)
        nu, P = self.nu, self.P
        unit = unit_bridge_magnitude(self.alpha)
        coef_only = 'coef' in init and 'global_scale' not in init
        if 'coef' in init:
            coef = np.array(init['coef'], dtype=np.float64)
        else:
            coef = np.zeros(P)
            if self.nu > 0:
                if self.family == 'linear':
                    coef[0] = self.outcome.mean()
                else:
                    ph = self.outcome[0].mean() / self.outcome[1].mean()
                    coef[0] = np.log(ph / (1 - ph))
        # bayesbridge.py:355-370
        if 'obs_prec' in init:
            obs_prec = np.array(init['obs_prec'], dtype=np.float64)
        elif self.family == 'linear':
            obs_prec = np.mean(
                (self.outcome - self.design.dot(coef)) ** 2) ** -1
        else:
            obs_prec = pg_mean(self.outcome[1], self.design.dot(coef))
        if coef_only:
            beta = coef[nu:]
            phi = len(beta) / self.alpha / np.sum(np.abs(beta) ** self.alpha)
            gscale = max(phi ** -(1 / self.alpha), .001 / unit)
            lscale = self.draw_lscale(gscale, beta)
        else:
            gscale = float(init['global_scale'])
            lscale = np.array(init.get('local_scale', np.ones(P - nu)),
                              dtype=np.float64)
        if gscale_parametrization == 'coef_magnitude':   # prior.py:129-141
            gscale = gscale / unit
            lscale = lscale * unit
        optim = None
        if 'coef' not in init:
            coef, optim = search_mode(
                self.family, self.design, self.outcome, coef, lscale, gscale,
                obs_prec, self.sd_unshrunk, self.slab)
            obs_prec = self.draw_obs_prec(coef)
            lscale = self.draw_lscale(gscale, coef[nu:])
        summ = CoefSummarizer(P, nu, self.slab)
        out = {'coef': np.zeros((P, n_iter)), 'global_scale': np.zeros(n_iter),
               'local_scale': np.zeros((P - nu, n_iter)),
               'logp': np.zeros(n_iter), 'n_cg_iter': np.zeros(n_iter)}
        records = [] if record else None
        for it in range(n_iter):
            coef, info = self.draw_coef(obs_prec, gscale, lscale, summ,
                                        records)
            obs_prec = self.draw_obs_prec(coef)
            gscale = self.draw_gscale(coef[nu:])
            lscale = self.draw_lscale(gscale, coef[nu:])
            out['coef'][:, it] = coef
            out['global_scale'][it] = gscale
            out['local_scale'][:, it] = lscale
            out['logp'][it] = self.logp(coef, gscale, obs_prec)
            out['n_cg_iter'][it] = info['n_iter']
        if gscale_parametrization == 'coef_magnitude':
            out['global_scale'] *= unit
            out['local_scale'] /= unit
        out['records'] = records
        out['init_optim'] = optim
        return out
