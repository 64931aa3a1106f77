"""Caller of the CG sampler: the 'cg' branch of the reference's
`SparseRegressionCoefficientSampler` (reg_coef_sampler/reg_coef_sampler.py:
60-103,194-201), its running posterior summaries
(reg_coef_posterior_summarizer.py:3-41,68-124) and the one-off initial mode
search (reg_coef_sampler.py:281-391) -- host-side NumPy around HIP products."""
import numpy as np
import scipy.optimize

from .cg_sampler import HipCGSampler


def compute_prior_shrunk_scale(gscale, lscale, slab_size):
    """tau*lambda / sqrt(1 + (tau*lambda/slab)^2) (reg_coef_sampler.py:194-201)."""
    scale = gscale * lscale
    return scale / np.sqrt(1 + (scale / slab_size) ** 2)


class OntheflySummarizer():
    """Running mean and second moment (reg_coef_posterior_summarizer.py:68-124)."""

    def __init__(self, n_param, sd_prior_samplesize=5):
        self.sd_prior_samplesize = sd_prior_samplesize
        self.sd_prior_guess = np.ones(n_param)
        self.n_averaged = 0
        self.stats = {'mean': np.zeros(n_param), 'square': np.ones(n_param)}

    def update_stats(self, theta):
        w = 1 / (1 + self.n_averaged)
        self.stats['mean'] = w * theta + (1 - w) * self.stats['mean']
        self.stats['square'] = w * theta ** 2 + (1 - w) * self.stats['square']
        self.n_averaged += 1

    def estimate_post_sd(self):
        if self.n_averaged <= 1:
            return self.sd_prior_guess
        k = self.n_averaged
        var = k / (k - 1) * (self.stats['square'] - self.stats['mean'] ** 2)
        w = (k - 1) / (k - 1 + self.sd_prior_samplesize)
        return np.sqrt(w * var + (1 - w) * self.sd_prior_guess ** 2)


class RegressionCoeffficientPosteriorSummarizer():
    """reg_coef_posterior_summarizer.py:3-41 (the spelling is the reference's)."""

    def __init__(self, n_coef, n_unshrunk, regularizing_slab_size):
        self.n_unshrunk = n_unshrunk
        self.slab_size = regularizing_slab_size
        self.coef_scaled_summarizer = OntheflySummarizer(n_coef)

    def update(self, coef, gscale, lscale):
        coef_scaled = coef.copy()
        coef_scaled[self.n_unshrunk:] /= compute_prior_shrunk_scale(
            gscale, lscale, self.slab_size)
        self.coef_scaled_summarizer.update_stats(coef_scaled)

    def extrapolate_coef_condmean(self, gscale, lscale):
        guess = self.coef_scaled_summarizer.stats['mean'].copy()
        guess[self.n_unshrunk:] *= compute_prior_shrunk_scale(
            gscale, lscale, self.slab_size)
        return guess

    def estimate_coef_precond_scale_sd(self):
        return self.coef_scaled_summarizer.estimate_post_sd()


class HipRegressionCoefficientSampler():

    def __init__(self, n_coef, prior_sd_for_unshrunk, sampling_method='cg',
                 regularizing_slab_size=float('inf')):
        if sampling_method != 'cg':
            raise ValueError("Only 'cg' sampler supported with HIP matrices.")
        self.prior_sd_for_unshrunk = np.asarray(prior_sd_for_unshrunk,
                                                dtype=np.float64)
        self.n_unshrunk = len(self.prior_sd_for_unshrunk)
        self.regularizing_slab_size = regularizing_slab_size
        self.regcoef_summarizer = RegressionCoeffficientPosteriorSummarizer(
            n_coef, self.n_unshrunk, regularizing_slab_size)
        self.cg_sampler = HipCGSampler(self.n_unshrunk)

    def get_internal_state(self):
        return {'regcoef_summarizer': self.regcoef_summarizer}

    def set_internal_state(self, state):
        self.regcoef_summarizer = state['regcoef_summarizer']

    def sample_gaussian_posterior(self, y, design, obs_prec, gscale, lscale,
                                  method='cg'):
        """reg_coef_sampler.py:60-103, 'cg' branch."""
        if method != 'cg':
            raise NotImplementedError()
        v = design.Tdot(obs_prec * y)                                    # :74
        prior_sd = np.concatenate((
            self.prior_sd_for_unshrunk,
            compute_prior_shrunk_scale(gscale, lscale,
                                       self.regularizing_slab_size)))
        with np.errstate(divide='ignore'):
            prior_prec_sqrt = 1 / prior_sd                               # :79
        guess = self.regcoef_summarizer.extrapolate_coef_condmean(
            gscale, lscale)
        sd = self.regcoef_summarizer.estimate_coef_precond_scale_sd()
        coef, cg_info = self.cg_sampler.sample(
            design, obs_prec, prior_prec_sqrt, v, coef_cg_init=guess,
            precond_by='prior', coef_scaled_sd=sd, maxiter=500,
            atol=10e-6 * np.sqrt(design.shape[1]))                       # :90-96
        self.regcoef_summarizer.update(coef, gscale, lscale)
        return coef, {'n_cg_iter': cg_info['n_iter']}

    def search_mode(self, coef, lscale, gscale, obs_prec, model,
                    optim_maxiter=250):
        """Conditional posterior mode of the coefficients by L-BFGS-B in
        prior-preconditioned coordinates (reg_coef_sampler.py:281-391 with the
        default, non-Newton options: maxcor 200, gtol 1e-6/sqrt(P))."""
        n_coef = len(coef)
        nu = self.n_unshrunk
        scale = np.ones(n_coef)
        scale[nu:] = compute_prior_shrunk_scale(
            gscale, lscale, self.regularizing_slab_size)
        with np.errstate(divide='ignore'):
            prior_prec = np.concatenate((
                (self.prior_sd_for_unshrunk / scale[:nu]) ** -2,
                np.ones(n_coef - nu)))
        design = model.design

        def loglik_and_grad(beta, loglik_only):
            if model.name == 'linear':
                return model.compute_loglik_and_gradient(
                    beta, obs_prec, loglik_only=loglik_only)
            return model.compute_loglik_and_gradient(
                beta, loglik_only=loglik_only)

        def neg_logp(theta):
            logp, _ = loglik_and_grad(theta * scale, True)
            return -(logp + np.sum(-prior_prec * theta ** 2) / 2)

        def neg_grad(theta):
            logp, grad = loglik_and_grad(theta * scale, False)
            return -(scale * grad - prior_prec * theta)

        design.memoize_dot(True)
        design.reset_matvec_count()
        result = scipy.optimize.minimize(
            neg_logp, coef / scale, method='L-BFGS-B', jac=neg_grad,
            options={'maxiter': optim_maxiter,
                     'gtol': 10 ** -6 / np.sqrt(n_coef), 'maxcor': 200})
        design.memoize_dot(False)
        info = {'is_success': result.success, 'method': 'L-BFGS-B',
                'n_iter': result['nit'], 'n_logp_eval': result['nfev'],
                'n_grad_eval': result.get('njev', 0),
                'n_design_matvec': design.n_matvec}
        return scale * result.x, info
