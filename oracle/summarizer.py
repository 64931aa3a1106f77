"""Oracle restatement of the running posterior summaries that feed the CG
warm start and the preconditioner (test infrastructure).

Follows reg_coef_sampler/reg_coef_posterior_summarizer.py:3-41 (scaling by the
regularised prior scale) and :68-124 (on-the-fly mean / second moment, the
shrunk sd estimate with a prior sample size of 5)."""
import numpy as np


def regularized_prior_scale(gscale, lscale, slab_size):
    """reg_coef_sampler.py:194-201 / reg_coef_posterior_summarizer.py:37-41."""
    raw = gscale * lscale
    return raw / np.sqrt(1.0 + (raw / slab_size) ** 2)


class CoefSummarizer:

    def __init__(self, n_coef, n_unshrunk, slab_size, sd_prior_samplesize=5):
        self.n_unshrunk = n_unshrunk
        self.slab_size = slab_size
        self.sd_prior_samplesize = sd_prior_samplesize
        self.n_averaged = 0
        self.mean = np.zeros(n_coef)        # :88-91 initial mean 0
        self.square = np.ones(n_coef)       # :88-91 initial 2nd moment 1

    def _scaled(self, coef, gscale, lscale):
        out = np.array(coef, dtype=np.float64, copy=True)
        out[self.n_unshrunk:] /= regularized_prior_scale(
            gscale, lscale, self.slab_size)
        return out

    def update(self, coef, gscale, lscale):
        theta = self._scaled(coef, gscale, lscale)
        w = 1.0 / (1.0 + self.n_averaged)                # :95
        self.mean = w * theta + (1.0 - w) * self.mean
        self.square = w * theta ** 2 + (1.0 - w) * self.square
        self.n_averaged += 1

    def extrapolate_coef_condmean(self, gscale, lscale):
        guess = self.mean.copy()                          # :25-29
        guess[self.n_unshrunk:] *= regularized_prior_scale(
            gscale, lscale, self.slab_size)
        return guess

    def estimate_post_sd(self):
        if self.n_averaged > 1:                           # :111-121
            k = self.n_averaged
            var = k / (k - 1) * (self.square - self.mean ** 2)
            w = (k - 1) / (k - 1 + self.sd_prior_samplesize)
            return np.sqrt(w * var + (1.0 - w) * 1.0)
        return np.ones_like(self.mean)

    def get_state(self):
        return {'mean': self.mean.copy(), 'square': self.square.copy(),
                'n_averaged': self.n_averaged}

    def set_state(self, state):
        self.mean = np.array(state['mean'], dtype=np.float64)
        self.square = np.array(state['square'], dtype=np.float64)
        self.n_averaged = int(state['n_averaged'])
