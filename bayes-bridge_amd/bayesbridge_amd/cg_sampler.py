"""The CG sampler behind the reference's ConjugateGradientSampler interface.

`HipCGSampler(n_coef_wo_shrinkage).sample(design, obs_prec, prior_prec_sqrt, z,
coef_cg_init, precond_by, coef_scaled_sd, maxiter, atol, seed)` has the
signature, return value `(coef, {'n_iter','valid_input','converged'})` and
warning behaviour of reg_coef_sampler/cg_sampler.py:15-94; the arithmetic runs
in libbbx.so on the MI355X (bbx_cg_sample).
"""
from ctypes import byref, c_int, c_void_p
from warnings import warn

import numpy as np

from . import _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class HipCGSampler():

    def __init__(self, n_coef_wo_shrinkage):
        self.n_coef_wo_shrinkage = n_coef_wo_shrinkage
        self._lib = _lib.load()

    def sample(
            self, design, obs_prec, prior_prec_sqrt, z,
            coef_cg_init=None, precond_by='prior', coef_scaled_sd=None,
            maxiter=None, atol=10e-6, seed=None, device_rng_seed=None):
        """See cg_sampler.py:20-41.  `device_rng_seed`: if given, the two
        Gaussian vectors are drawn on the GPU (Philox) instead of from the
        global NumPy stream -- distribution parity only."""
        if precond_by != 'prior':
            # cg_sampler.py:140-149: 'diag'/None are never used by the Gibbs
            # path; they need compute_fisher_info (outside the hot path).
            raise NotImplementedError(
                "only precond_by='prior' is implemented on the HIP path")
        if not getattr(design, 'use_hip', False):
            raise TypeError("design must be a HipDesignMatrix")
        n, P = design.shape
        if seed is not None:
            np.random.seed(seed)  # cg_sampler.py:51-52
        obs_prec = np.ascontiguousarray(
            np.broadcast_to(np.asarray(obs_prec, dtype=np.float64), (n,)))
        prior_prec_sqrt = np.ascontiguousarray(prior_prec_sqrt, np.float64)
        z = np.ascontiguousarray(z, np.float64)
        if coef_cg_init is None:
            coef_cg_init = np.zeros(P)
        if coef_scaled_sd is None:
            coef_scaled_sd = np.ones(P)
        x0 = np.ascontiguousarray(coef_cg_init, np.float64)
        sd = np.ascontiguousarray(coef_scaled_sd, np.float64)
        for name, a in (('prior_prec_sqrt', prior_prec_sqrt), ('z', z),
                        ('coef_cg_init', x0), ('coef_scaled_sd', sd)):
            if a.shape != (P,):
                raise ValueError("%s must have length %d" % (name, P))
        if maxiter is None:
            maxiter = 10 * P  # SciPy's default (cg: maxiter = n*10)
        if device_rng_seed is None:
            # Draw the target vector exactly as the reference does
            # (cg_sampler.py:61-62: global NumPy stream, n first, then P).
            randn_vec_1 = np.random.randn(n)
            randn_vec_2 = np.random.randn(P)
            dev_seed = 0
        else:
            randn_vec_1 = randn_vec_2 = None
            dev_seed = int(device_rng_seed)
        coef = np.empty(P, dtype=np.float64)
        n_iter, info = c_int(0), c_int(0)
        _lib.check(self._lib.bbx_cg_sample(
            design.handle, _ptr(obs_prec), _ptr(prior_prec_sqrt), _ptr(z),
            _ptr(x0), _ptr(sd), int(self.n_coef_wo_shrinkage),
            _ptr(randn_vec_1), _ptr(randn_vec_2), dev_seed, int(maxiter),
            float(atol), _ptr(coef), byref(n_iter), byref(info)))
        info = info.value
        if info != 0:
            warn(
                "The conjugate gradient algorithm did not achieve the requested " +
                "tolerance level. You may increase the maxiter or use the dense " +
                "linear algebra instead."
            )  # cg_sampler.py:82-87
        cg_info = {'n_iter': n_iter.value}
        cg_info['valid_input'] = (info >= 0)
        cg_info['converged'] = (info == 0)
        return coef, cg_info
