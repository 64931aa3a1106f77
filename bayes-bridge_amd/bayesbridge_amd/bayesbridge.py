"""Gibbs driver for the 'cg' path on an MI355X.

`BayesBridge(model, prior).gibbs(n_iter, n_burnin, thin, seed, init,
params_to_save, coef_sampler_type, n_status_update, options)` has the
signature, the output dictionaries and the update order of the reference
driver (bayesbridge.py:109-277: beta -> Omega -> tau -> lambda -> logp ->
store).  Two execution modes, chosen by `options={'rng': ...}`:

  'device' (default)  the whole iteration stays in HBM (libbbx bbx_chain_run);
                      random numbers are Philox counters on the GPU.  Same
                      distribution as the reference, different stream.
  'reference'         the reference's own GPU arrangement (its CuPy mode,
                      SURVEY.md 3.4): only the CG solve runs on the GPU, every
                      random number comes from the same host streams in the
                      same order as the reference (global NumPy MT19937 for
                      eta and the Gamma draws; two PCG64 generators for
                      Polya-Gamma and tilted stable, random/random.py:17-22),
                      so a chain reproduces the reference's samples for the
                      same seed to floating-point tolerance.
"""
import math
import time
from ctypes import c_void_p
from warnings import warn

import numpy as np

from . import _lib
from .device_chain import HipGibbsChain
from .hostrng import ReferenceRandom
from .model import LogisticModel
from .prior import RegressionCoefPrior, unit_magnitude
from .reg_coef_sampler import (HipRegressionCoefficientSampler,
                               RegressionCoeffficientPosteriorSummarizer)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class SamplerOptions():
    """gibbs_util.py:7-84 restricted to what a HIP design supports."""

    def __init__(self, coef_sampler_type='cg', global_scale_update='sample',
                 hmc_curvature_est_stabilized=False, rng='device'):
        if coef_sampler_type not in ('cholesky', 'cg', 'hmc'):
            raise ValueError("Unsupported regression coefficient sampler.")
        if coef_sampler_type != 'cg':
            # gibbs_util.py:49-50 says the same about CuPy matrices
            raise ValueError("Only 'cg' sampler supported with HIP matrices.")
        if rng not in ('device', 'reference'):
            raise ValueError("rng must be 'device' or 'reference'")
        if global_scale_update not in ('sample', 'optimize', None):
            raise ValueError("Unsupported global scale update.")
        self.coef_sampler_type = coef_sampler_type
        self.gscale_update = global_scale_update
        self.curvature_est_stabilized = hmc_curvature_est_stabilized
        self.rng = rng

    def get_info(self):
        return {
            'coef_sampler_type': self.coef_sampler_type,
            'global_scale_update': self.gscale_update,
            'hmc_curvature_est_stabilized': self.curvature_est_stabilized,
            'rng': self.rng,
        }

    @staticmethod
    def pick_default_and_create(coef_sampler_type, options, model_name, design):
        options = dict(options) if options else {}
        if 'coef_sampler_type' in options:
            if coef_sampler_type is not None:
                warn("Duplicate specification of method for sampling "
                     "regression coefficient. Will use the dictionary one.")
            coef_sampler_type = options['coef_sampler_type']
        if coef_sampler_type not in (None, 'cholesky', 'cg', 'hmc'):
            raise ValueError("Unsupported sampler type.")
        if coef_sampler_type not in (None, 'cg'):
            raise ValueError("Only 'cg' sampler supported with HIP matrices.")
        if model_name not in ('linear', 'logit'):
            raise ValueError("Only linear and logit models use the CG sampler.")
        n_obs, n_pred = design.shape
        if n_pred > n_obs:
            warn("Sampler has not been optimized for 'small n' problem.")
        options['coef_sampler_type'] = 'cg'
        return SamplerOptions(**options)


class BayesBridge():
    """Gibbs sampler for Bayesian bridge sparse regression (bayesbridge.py:12)."""

    def __init__(self, model, prior=None):
        if prior is None:
            prior = RegressionCoefPrior()
        if not getattr(model.design, 'use_hip', False):
            raise TypeError("model.design must be a HipDesignMatrix; build the "
                            "model with bayesbridge_amd.RegressionModel")
        self.n_obs = model.n_obs
        self.n_pred = model.n_pred
        self.n_unshrunk = prior.n_fixed
        self.prior_sd_for_unshrunk = prior.sd_for_fixed.copy()
        if model.intercept_added:                       # bayesbridge.py:26-32
            self.n_unshrunk += 1
            self.prior_sd_for_unshrunk = np.concatenate((
                [prior.sd_for_intercept], self.prior_sd_for_unshrunk))
        self.prior_sd_for_unshrunk = np.asarray(
            self.prior_sd_for_unshrunk, dtype=np.float64)
        self.model = model
        self.prior = prior
        self.rg = None
        self._lib = _lib.load()
        self._chain = None
        self._chain_seed = None

    def __del__(self):
        self._destroy_chain()

    def _destroy_chain(self):
        ch = getattr(self, '_chain', None)
        if ch is not None:
            ch.close()
        self._chain = None

    # ------------------------------------------------------------------ API
    def gibbs_resume(self, prev_mcmc_info, n_add_iter, n_status_update=0,
                     merge=False, prev_samples=None):
        """Continue a chain from `mcmc_info` of a previous gibbs call
        (bayesbridge.py:43-107)."""
        if merge and prev_samples is None:
            raise ValueError(
                "To merge the outputs from previous and new MCMC runs, you "
                "have to supply the optional argument `prev_samples`.")
        init = prev_mcmc_info['_markov_chain_state']
        if '_markov_chain_state_raw' in prev_mcmc_info:
            init = {'_raw': prev_mcmc_info['_markov_chain_state_raw']}
        new_samples, new_mcmc_info = self.gibbs(
            n_add_iter, 0, prev_mcmc_info['thin'], init=init,
            params_to_save=prev_mcmc_info['saved_params'],
            n_status_update=n_status_update,
            options=prev_mcmc_info['options'], _resume_from=prev_mcmc_info)
        if merge:
            new_samples = {
                key: np.concatenate((prev_samples[key], new_samples[key]),
                                    axis=-1) for key in new_samples}
            prev_info = prev_mcmc_info['_reg_coef_sampling_info']
            next_info = new_mcmc_info['_reg_coef_sampling_info']
            new_mcmc_info['_reg_coef_sampling_info'] = {
                key: np.concatenate((prev_info[key], next_info[key]), axis=-1)
                for key in prev_info}
            new_mcmc_info['n_iter'] += prev_mcmc_info['n_iter']
            new_mcmc_info['runtime'] += prev_mcmc_info['runtime']
            for key in ('_init_optim_info', 'seed'):
                new_mcmc_info[key] = prev_mcmc_info[key]
        return new_samples, new_mcmc_info

    def gibbs(self, n_iter, n_burnin=0, thin=1, seed=None,
              init={'global_scale': 0.1},
              params_to_save=('coef', 'global_scale', 'logp'),
              coef_sampler_type=None, n_status_update=0, options=None,
              _resume_from=None, _device_out=None):
        """Generate posterior samples (bayesbridge.py:109-277).  Returns
        (samples, mcmc_info); `samples['coef']` is (n_pred, n_sample) with the
        MCMC index last.

        _device_out (device-RNG mode; chains.run_chains): {name: torch CUDA
        tensor, float64, contiguous, SAMPLE-major [n_sample, dim]} for names
        out of 'coef', 'local_scale', 'obs_prec' -- those samples are left in
        HBM (ready for the RCCL gather) and are absent from `samples`."""
        if not isinstance(options, SamplerOptions):
            options = SamplerOptions.pick_default_and_create(
                coef_sampler_type, options, self.model.name, self.model.design)
        if params_to_save == 'all':
            params_to_save = ('coef', 'local_scale', 'global_scale', 'logp',
                              'obs_prec')
        start_time = time.time()
        if options.rng == 'reference':
            if _device_out:
                raise ValueError("_device_out needs the device-RNG mode")
            out = self._gibbs_reference_rng(
                n_iter, n_burnin, thin, seed, init, params_to_save, options,
                n_status_update, _resume_from)
        else:
            out = self._gibbs_device(
                n_iter, n_burnin, thin, seed, init, params_to_save, options,
                _resume_from, _device_out, n_status_update)
        samples, mcmc_info = self._package(
            out, n_iter, n_burnin, thin, seed, params_to_save, options,
            time.time() - start_time)
        if _device_out and 'local_scale' in _device_out \
                and self.prior._gscale_paramet == 'coef_magnitude':
            # the rescaling of bayesbridge.py:244-251, where the samples are
            # (a true division, as NumPy's on the host samples: torch turns
            # division by a Python scalar into a multiplication by 1 / m)
            t = _device_out['local_scale']
            t.div_(t.new_full((1,), unit_magnitude(self.prior.bridge_exp)))
        return samples, mcmc_info

    # what chains.run_chains looks for before it hands out device buffers
    supports_device_samples = True

    def _package(self, out, n_iter, n_burnin, thin, seed, params_to_save,
                 options, runtime):
        """(samples, mcmc_info) of one chain in the reference's format
        (bayesbridge.py:242-277)."""
        samples, sampling_info, state, init_used, optim_info, extra = out
        coef, obs_prec, lscale, gscale = state
        # exact (raw-parametrisation) state: resuming from it is bitwise
        # equivalent to not having stopped; the user-facing state below goes
        # through the 'coef_magnitude' rescaling and back, which is not.
        raw_state = {'coef': np.array(coef, copy=True),
                     'obs_prec': np.array(obs_prec, copy=True),
                     'local_scale': np.array(lscale, copy=True),
                     'global_scale': float(gscale)}
        if self.prior._gscale_paramet == 'coef_magnitude':  # bayesbridge.py:244-251
            gscale, lscale = self.prior.adjust_scale(
                gscale, lscale, to='coef_magnitude')
            self.prior.adjust_scale(
                samples.get('global_scale', 0.), samples.get('local_scale', 0.),
                to='coef_magnitude')
        mcmc_info = {
            'init': init_used,
            'n_iter': n_iter,
            'n_burnin': n_burnin,
            'thin': thin,
            'seed': seed,
            'n_coef_wo_shrinkage': self.n_unshrunk,
            'prior_sd_for_unshrunk': self.prior_sd_for_unshrunk,
            'bridge_exponent': self.prior.bridge_exp,
            'coef_sampler_type': options.coef_sampler_type,
            'saved_params': params_to_save,
            'runtime': runtime,
            'options': options.get_info(),
            '_init_optim_info': optim_info,
            '_reg_coef_sampling_info': sampling_info,
            '_markov_chain_state': {
                'coef': coef, 'local_scale': lscale, 'global_scale': gscale,
                'obs_prec': obs_prec},
            '_markov_chain_state_raw': raw_state,
        }
        mcmc_info.update(extra)
        return samples, mcmc_info

    def batch_width(self, n_chain, params_to_save=('coef', 'global_scale',
                                                   'logp'), options=None):
        """How many of `n_chain` chains one batch can hold on this model's
        design (0: batching does not apply or does not pay): sparse tiled
        designs of binary covariates (plus, possibly, dense continuous columns)
        4 while they are small, else 2 (designs with other stored
        values can be batched explicitly -- HipChainBatch -- but run faster one
        chain at a time), dense
        designs (f32 or f64 storage) 4, 8, 16 or 32 (the batched dense products read the matrix twice
        per operator application whatever the width -- 2.3 single-chain
        applications, 2.9 at 32 chains: two chains run faster one after the
        other).  Batches keep 'coef', 'global_scale', 'logp' and use
        the device RNG."""
        design = self.model.design
        if options is not None and not isinstance(options, SamplerOptions):
            options = SamplerOptions.pick_default_and_create(
                None, options, self.model.name, design)
        if options is not None and options.rng != 'device':
            return 0
        if params_to_save == 'all' or not set(params_to_save) <= {
                'coef', 'global_scale', 'logp'}:
            return 0
        if design.is_sparse:
            if design.storage_format != 'tiled':
                return 0
            hy = design.hybrid_info
            if not design.is_binary and hy is None:
                # stored values throughout: the batch would go through the
                # plain valued K-layout, slower than two chains one after the
                # other.  (Mixed designs keep their split layout in a batch:
                # value-free K-layout + dense block + valued rest.)
                return 0
            valued_rest = hy is not None and hy['rest_nnz'] > 0
            # four chains per pass pay while the design is small (fixed costs
            # per launch dominate: 1.4-1.9x at 5k x 500 ... 100k x 10k against
            # 1.1-1.5x for pairs); at 1M x 50k the four planes shrink the LDS
            # tiles too far (0.99x against 1.32x; profiles/r03_small_batches.txt).
            # The library's cost model decides (bbx_batch_predict: 0.67 for
            # four chains at 1M x 50k, 2.6 at 100k x 10k); HipChainBatch
            # refuses what it prices below 1.
            widths = (2,) if valued_rest else (4, 2)   # valued kernels: pairs only
        else:
            # the first batch on a dense design builds a transposed copy of
            # the matrix: it has to fit next to the matrix itself
            try:
                import torch
                free, _ = torch.cuda.mem_get_info(design.device)
                n_, P_ = design.shape
                el = 4 if design.storage_dtype == 'float32' else 8
                have_copy = design.storage_bytes > 1.9 * n_ * P_ * el
                if not have_copy and free < 1.1 * n_ * P_ * el + (1 << 30):
                    return 0
            except Exception:
                return 0     # no memory figure: do not risk the transposed copy
            widths = (32, 16, 8, 4, 2)
        from .device_chain import HipChainBatch
        for w in widths:
            if w <= n_chain and HipChainBatch.predicted_speedup(design, w) > 1.:
                return w
        return 0

    def gibbs_batch(self, seeds, n_iter, n_burnin=0, thin=1,
                    init={'global_scale': 0.1},
                    params_to_save=('coef', 'global_scale', 'logp'),
                    options=None, allow_slow=False):
        """len(seeds) chains of `gibbs(...)` stepped as ONE batch that shares
        every pass over the design (HipChainBatch; csrc/batch.hip).  The
        reference's way to more chains is more processes (bayesbridge.py:109).
        Returns a list of (samples, mcmc_info), one per seed, in gibbs()'s
        format; a chain's samples do not depend on its companions (they do
        depend on the batch WIDTH, to rounding).  A width the cost model prices
        below single chains raises unless `allow_slow`."""
        import copy
        from .device_chain import HipChainBatch
        if not isinstance(options, SamplerOptions):
            options = SamplerOptions.pick_default_and_create(
                None, options, self.model.name, self.model.design)
        if options.rng != 'device':
            raise ValueError("batched chains use the device RNG")
        if not set(params_to_save) <= {'coef', 'global_scale', 'logp'}:
            raise ValueError("a batch keeps 'coef', 'global_scale' and 'logp'")
        start_time = time.time()
        seeds = [int(sd) for sd in seeds]
        chains, setups = [], []
        batch = None
        try:
            for sd in seeds:
                chain = self._new_chain(sd)
                chains.append(chain)
                setups.append(self._device_setup(chain, sd,
                                                 copy.deepcopy(init), options))
            batch = HipChainBatch(chains, allow_slow=allow_slow)
            kept, _ = batch.run(n_iter, n_burnin, thin, maxiter=500, atol=0.,
                                save_coef='coef' in params_to_save)
            n_unconv = batch.n_unconverged       # per chain
            runtime = time.time() - start_time
            results = []
            for k, (chain, sd) in enumerate(zip(chains, seeds)):
                mine = {key: val[k] for key, val in kept.items()}
                out = self._device_collect(chain, sd, mine, n_unconv[k],
                                           n_iter, n_burnin, thin,
                                           params_to_save, *setups[k])
                results.append(self._package(out, n_iter, n_burnin, thin, sd,
                                             params_to_save, options, runtime))
                results[-1][1]['batch'] = {'width': len(seeds), 'slot': k}
            return results
        finally:
            # an error inside the run (BBX_ERR_NUMERIC, out of memory while the
            # dense transposed copy is built) must not leave up to 32 chains'
            # device buffers to the garbage collector
            if batch is not None:
                batch.close()
            for chain in chains:
                chain.close()

    def gibbs_multichain(self, n_chain, n_iter, n_burnin=0, thin=1, seed=0,
                         init={'global_scale': 0.1},
                         params_to_save=('coef', 'global_scale', 'logp'),
                         options=None, batch=False):
        """`n_chain` independent chains, seeds seed + k; under
        torch.distributed.run one rank per GPU shares them and the samples are
        gathered once over RCCL on rank 0 (`chains.run_chains`).  The
        reference has one chain per process (bayesbridge.py:109); this is the
        multi-GPU capability of SURVEY.md 8(e).  Returns (samples, infos) with
        samples[name] shaped (n_chain, ..., n_sample) on rank 0.  `batch`:
        False (default; chain k's samples do not depend on the number of
        ranks), 'auto' or a width -- see chains.run_chains."""
        from . import chains
        return chains.run_chains(self, n_chain, n_iter, n_burnin, thin, seed,
                                 init, params_to_save, options, batch=batch)

    # ------------------------------------------------- shared initialisation
    def _pre_allocate(self, n_post_burnin, thin, params_to_save):
        n_sample = math.floor(n_post_burnin / thin)        # gibbs_util.py:122
        samples = {}
        if 'coef' in params_to_save:
            samples['coef'] = np.zeros((self.n_pred, n_sample))
        if 'local_scale' in params_to_save:
            samples['local_scale'] = np.zeros(
                (self.n_pred - self.n_unshrunk, n_sample))
        if 'global_scale' in params_to_save:
            samples['global_scale'] = np.zeros(n_sample)
        if 'obs_prec' in params_to_save:
            if self.model.name == 'linear':
                samples['obs_prec'] = np.zeros(n_sample)
            else:
                samples['obs_prec'] = np.zeros((self.n_obs, n_sample))
        if 'logp' in params_to_save:
            samples['logp'] = np.zeros(n_sample)
        return samples, {'n_cg_iter': np.zeros(n_sample)}

    def _initial_obs_prec(self, init, coef):
        if 'obs_prec' in init:                          # bayesbridge.py:355-370
            obs_prec = np.array(init['obs_prec'], dtype=np.float64, copy=True,
                                order='C')
            expected = self.n_obs if self.model.name == 'logit' else 1
            if obs_prec.size != expected:
                raise ValueError('An invalid initial state.')
            return obs_prec if self.model.name == 'logit' else float(obs_prec)
        if self.model.name == 'linear':
            return np.mean(
                (self.model.y - self.model.design.dot(coef)) ** 2) ** -1
        return LogisticModel.compute_polya_gamma_mean(
            self.model.n_trial, self.model.design.dot(coef))

    def _initialize_chain(self, init, bridge_exp, update_local_scale,
                          update_obs_precision, update_global_scale,
                          sampler):
        """initialize_chain (bayesbridge.py:279-353), parameterised by the
        update functions of the active mode."""
        if isinstance(init, dict) and '_raw' in init:
            raw = init['_raw']
            obs = raw['obs_prec']
            obs_prec = np.array(obs, dtype=np.float64, copy=True) \
                if np.ndim(obs) > 0 else float(obs)
            coef = np.array(raw['coef'], dtype=np.float64, copy=True)
            lscale = np.array(raw['local_scale'], dtype=np.float64, copy=True)
            gscale = float(raw['global_scale'])
            return (coef, obs_prec, lscale, gscale,
                    {'coef': coef, 'obs_prec': obs_prec,
                     'local_scale': lscale, 'global_scale': gscale}, None)
        for key in init:
            if key not in ('coef', 'local_scale', 'global_scale', 'obs_prec',
                           'logp'):
                warn("'{:s}' is not a valid parameter name and "
                     "will be ignored.".format(key))
        coef_only_specified = 'coef' in init and ('global_scale' not in init)
        if 'coef' in init:
            coef = np.array(init['coef'], dtype=np.float64, copy=True)
            if not len(coef) == self.n_pred:
                raise ValueError(
                    'Invalid initial length of regression coefficient.')
        else:
            coef = np.zeros(self.n_pred)
            if self.model.intercept_added:
                coef[0] = self.model.calc_intercept_mle()
        obs_prec = self._initial_obs_prec(init, coef)
        n_shrunk = self.n_pred - self.n_unshrunk
        if coef_only_specified:
            gscale = update_global_scale(
                None, coef[self.n_unshrunk:], bridge_exp, method='optimize')
            lscale = update_local_scale(
                gscale, coef[self.n_unshrunk:], bridge_exp)
        else:
            if 'global_scale' not in init:
                raise ValueError("Initial global scale must be specified when "
                                 "coefficients aren't specified.")
            if self.prior._gscale_paramet == 'raw':
                warn("Using the raw global scale parametrization; make sure "
                     "that the specified initial value is scaled accordingly.")
            gscale = float(init['global_scale'])
            if 'local_scale' in init:
                lscale = np.array(init['local_scale'], dtype=np.float64,
                                  copy=True)
                if not len(lscale) == n_shrunk:
                    raise ValueError(
                        'Invalid initial length of local scale parameter')
            else:
                lscale = np.ones(n_shrunk)
        if self.prior._gscale_paramet == 'coef_magnitude':
            gscale, lscale = self.prior.adjust_scale(gscale, lscale, to='raw')
        optim_info = None
        if 'coef' not in init:
            coef, info = sampler.search_mode(
                coef, lscale, gscale, obs_prec, self.model)
            obs_prec = update_obs_precision(coef)
            lscale = update_local_scale(
                gscale, coef[self.n_unshrunk:], bridge_exp)
            optim_info = {key: info[key]
                          for key in ('is_success', 'n_design_matvec', 'n_iter')}
        init_used = {'coef': coef, 'obs_prec': obs_prec,
                     'local_scale': lscale, 'global_scale': gscale}
        return coef, obs_prec, lscale, gscale, init_used, optim_info

    def _lower_bd(self, bridge_exp, magnitude=.001):
        return magnitude / self.prior.compute_power_exp_ave_magnitude(
            bridge_exp)                                  # bayesbridge.py:420-424

    @staticmethod
    def _monte_carlo_em_global_scale(coef_under_shrinkage, bridge_exp):
        phi = len(coef_under_shrinkage) / bridge_exp \
            / np.sum(np.abs(coef_under_shrinkage) ** bridge_exp)
        return phi ** - (1 / bridge_exp)                 # bayesbridge.py:450-456

    # ---------------------------------------------- mode 1: reference streams
    def _gibbs_reference_rng(self, n_iter, n_burnin, thin, seed, init,
                             params_to_save, options, n_status_update,
                             resume_from):
        model, prior = self.model, self.prior
        design = model.design
        bridge_exp = prior.bridge_exp
        if self.rg is None:
            self.rg = ReferenceRandom()
        if resume_from is None:
            self.rg.set_seed(seed)
            sampler = HipRegressionCoefficientSampler(
                self.n_pred, self.prior_sd_for_unshrunk, 'cg', prior.slab_size)
        else:
            self.rg.set_state(resume_from['_random_gen_state'])
            sampler = HipRegressionCoefficientSampler(
                self.n_pred, self.prior_sd_for_unshrunk, 'cg', prior.slab_size)
            sampler.set_internal_state(resume_from['_reg_coef_sampler_state'])
        rg = self.rg
        nu = self.n_unshrunk

        def update_obs_precision(coef):                  # bayesbridge.py:397-410
            if model.name == 'linear':
                resid = model.y - design.dot(coef)
                scale = np.sum(resid ** 2) / 2
                obs_var = scale / rg.np_random.gamma(self.n_obs / 2, 1)
                return 1 / obs_var
            return rg.polya_gamma(model.n_trial.astype(np.intc),
                                  design.dot(coef))

        def update_global_scale(gscale, beta, bridge_exp, method='sample'):
            if beta.size == 0:                           # bayesbridge.py:412-448
                return 1.
            if method == 'optimize':
                gscale = self._monte_carlo_em_global_scale(beta, bridge_exp)
            elif method == 'sample':
                if np.count_nonzero(beta) == 0:
                    gscale = 0
                else:
                    hyper = prior.param['gscale_neg_power']
                    shape = hyper['shape'] + beta.size / bridge_exp
                    rate = hyper['rate'] + np.sum(np.abs(beta) ** bridge_exp)
                    phi = rg.np_random.gamma(shape, scale=1 / rate)
                    gscale = 1 / phi ** (1 / bridge_exp)
            lower_bd = self._lower_bd(bridge_exp)
            if (method is not None) and gscale < lower_bd:
                gscale = lower_bd
                warn("The global shrinkage parameter update returned an "
                     "unreasonably small value. Returning a specified lower "
                     "bound value instead.")
            return gscale

        def update_local_scale(gscale, beta, bridge_exp):
            if bridge_exp == 2:                          # bayesbridge.py:458-478
                return .5 * np.ones(beta.size)
            lscale = np.sqrt(.5 / rg.tilted_stable(
                bridge_exp / 2, (beta / gscale) ** 2))
            if np.any(lscale == 0):
                warn("Local scale parameter under-flowed. Replacing with a "
                     "small number.")
                lscale[lscale == 0] = 10e-16
            elif np.any(np.isinf(lscale)):
                warn("Local scale parameter over-flowed. Replacing with a "
                     "large number.")
                lscale[np.isinf(lscale)] = 2.0 / gscale
            return lscale

        def compute_posterior_logprob(coef, gscale, obs_prec):
            if model.name == 'linear':                   # bayesbridge.py:480-511
                loglik, _ = model.compute_loglik_and_gradient(
                    coef, obs_prec, loglik_only=True)
            else:
                loglik, _ = model.compute_loglik_and_gradient(
                    coef, loglik_only=True)
            loglik += - .5 * np.sum((coef / prior.slab_size) ** 2)
            n_shrunk = len(coef) - nu
            prior_logp = - n_shrunk * math.log(gscale) \
                - np.sum(np.abs(coef[nu:] / gscale) ** bridge_exp)
            sd = self.prior_sd_for_unshrunk
            prior_logp += - 1 / 2 * np.sum((coef[:nu] / sd) ** 2)
            prior_logp += - np.sum(np.log(sd[sd < float('inf')]))
            hyper = prior.param['gscale_neg_power']
            prior_logp += (hyper['shape'] - 1.) * math.log(gscale) \
                - hyper['rate'] * gscale
            return loglik + prior_logp

        coef, obs_prec, lscale, gscale, init_used, optim_info = \
            self._initialize_chain(init, bridge_exp, update_local_scale,
                                   update_obs_precision, update_global_scale,
                                   sampler)
        samples, sampling_info = self._pre_allocate(
            n_iter - n_burnin, thin, params_to_save)
        n_status_update = min(n_iter, n_status_update)
        stamp = time.time()
        for mcmc_iter in range(1, n_iter + 1):
            # beta | rest (bayesbridge.py:372-395)
            if model.name == 'linear':
                y_gaussian = model.y
                omega = obs_prec * np.ones(self.n_obs)
            else:
                omega = obs_prec
                y_gaussian = (model.n_success - model.n_trial / 2) / obs_prec
            coef, info = sampler.sample_gaussian_posterior(
                y_gaussian, design, omega, gscale, lscale, 'cg')
            obs_prec = update_obs_precision(coef)
            gscale = update_global_scale(
                gscale, coef[nu:], bridge_exp, method=options.gscale_update)
            lscale = update_local_scale(gscale, coef[nu:], bridge_exp)
            logp = compute_posterior_logprob(coef, gscale, obs_prec)
            if mcmc_iter > n_burnin and (mcmc_iter - n_burnin) % thin == 0:
                idx = math.floor((mcmc_iter - n_burnin) / thin) - 1
                if 'coef' in samples:
                    samples['coef'][:, idx] = coef
                if 'local_scale' in samples:
                    samples['local_scale'][:, idx] = lscale
                if 'global_scale' in samples:
                    samples['global_scale'][idx] = gscale
                if 'obs_prec' in samples:
                    if model.name == 'linear':
                        samples['obs_prec'][idx] = obs_prec
                    else:
                        samples['obs_prec'][:, idx] = obs_prec
                if 'logp' in samples:
                    samples['logp'][idx] = logp
                sampling_info['n_cg_iter'][idx] = info['n_cg_iter']
            if n_status_update and \
                    mcmc_iter % int(n_iter / n_status_update) == 0:
                now = time.time()
                print("{:d} Gibbs iterations complete: {:.3g} minutes has "
                      "elasped since the last update.".format(
                          mcmc_iter, (now - stamp) / 60))
                stamp = now
        extra = {'_random_gen_state': rg.get_state(),
                 '_reg_coef_sampler_state': sampler.get_internal_state()}
        return (samples, sampling_info, (coef, obs_prec, lscale, gscale),
                init_used, optim_info, extra)

    # -------------------------------------------------- mode 2: device chain
    def _new_chain(self, seed):
        model, prior = self.model, self.prior
        hyper = prior.param['gscale_neg_power']
        return HipGibbsChain(
            model.design, model.name,
            model.y if model.name == 'linear' else model.n_success,
            n_trial=model.n_trial if model.name == 'logit' else None,
            sd_unshrunk=self.prior_sd_for_unshrunk,
            bridge_exponent=prior.bridge_exp, slab_size=prior.slab_size,
            gscale_shape=hyper['shape'], gscale_rate=hyper['rate'], seed=seed)

    def _make_chain(self, seed):
        self._destroy_chain()
        self._chain = self._new_chain(seed)
        self._chain_seed = seed

    def _gibbs_device(self, n_iter, n_burnin, thin, seed, init,
                      params_to_save, options, resume_from, device_out=None,
                      n_status_update=0):
        model, prior = self.model, self.prior
        bridge_exp = prior.bridge_exp
        if resume_from is not None:
            seed = resume_from['_random_gen_state']['seed']
        elif seed is None:
            seed = int(np.random.SeedSequence().generate_state(1)[0])
        if resume_from is None or self._chain is None:
            self._make_chain(seed)
        elif self._chain_seed != seed:
            # the handle last ran another seed's streams: a resumed chain has
            # to continue ITS Philox key
            self._chain.seed = seed
            self._chain_seed = seed
        chain = self._chain
        init_used, optim_info = self._device_setup(chain, seed, init, options,
                                                   resume_from)
        # status lines as the reference prints them (gibbs_util.py:214-238),
        # from the library's host loop through a callback
        n_status_update = min(n_iter, int(n_status_update or 0))
        if n_status_update > 0:
            stamp = [time.time()]

            def report(mcmc_iter):
                now = time.time()
                print("{:d} Gibbs iterations complete: {:.3g} minutes has "
                      "elasped since the last update.".format(
                          mcmc_iter, (now - stamp[0]) / 60))
                stamp[0] = now
            chain.set_progress(int(n_iter / n_status_update), report)
        else:
            chain.set_progress(0)
        device_out = dict(device_out or {})
        host_params = tuple(k for k in params_to_save if k not in device_out)
        samples, _ = self._pre_allocate(n_iter - n_burnin, thin, host_params)
        if device_out:
            n_sample = (n_iter - n_burnin) // thin
            dims = {'coef': self.n_pred,
                    'local_scale': self.n_pred - self.n_unshrunk,
                    'obs_prec': self.n_obs if model.name == 'logit' else 1}
            for name, t in device_out.items():
                want = (n_sample, dims[name])
                if name not in params_to_save or not t.is_cuda \
                        or not t.is_contiguous() or str(t.dtype) != \
                        'torch.float64' or t.numel() != want[0] * want[1] \
                        or t.device.index != model.design.device:
                    raise ValueError(
                        "_device_out[%r] must be a contiguous float64 tensor "
                        "of %d x %d on the design's device" % ((name,) + want))
            if any(k in samples for k in ('coef', 'local_scale', 'obs_prec')):
                raise ValueError("vector samples go either all to the device "
                                 "or all to the host")
            gs, lp, ncg, n_unconv = chain.run_device(
                n_iter, n_burnin=n_burnin, thin=thin, maxiter=500, atol=0.,
                **{'d_%s_ptr' % {'coef': 'coef', 'local_scale': 'lscale',
                                 'obs_prec': 'obs_prec'}[k]: t.data_ptr()
                   for k, t in device_out.items()})
            kept = {'global_scale': gs, 'logp': lp, 'n_cg_iter': ncg}
        else:
            kept, n_unconv = chain.run(
                n_iter, n_burnin, thin, maxiter=500, atol=0.,
                save=[k for k in ('coef', 'local_scale', 'obs_prec')
                      if k in samples])
        return self._device_collect(chain, seed, kept, n_unconv, n_iter,
                                    n_burnin, thin, host_params, init_used,
                                    optim_info)

    def _device_setup(self, chain, seed, init, options, resume_from=None):
        """Brings a device chain to its initial state (bayesbridge.py:279-333:
        initial values, optional mode search); returns (init_used,
        optim_info)."""
        model, prior = self.model, self.prior
        bridge_exp = prior.bridge_exp
        sampler = HipRegressionCoefficientSampler(
            self.n_pred, self.prior_sd_for_unshrunk, 'cg', prior.slab_size)
        chain.set_gscale_update(options.gscale_update)
        if resume_from is not None:
            st = resume_from['_reg_coef_sampler_state']
            chain.set_summary(st['mean'], st['square'], st['n_averaged'])
            chain.iteration = resume_from['_random_gen_state']['iteration']
        nu = self.n_unshrunk
        host_rng = np.random.default_rng(seed)
        dev = model.design.device

        # Initialisation draws (once, O(n) / O(p)) reuse the device samplers
        # through the chain: set the state, then let one kernel refresh it.
        def update_obs_precision(coef):
            # bayesbridge.py:397-410 with the device samplers
            eta = model.design.dot(coef)
            if model.name == 'linear':
                scale = np.sum((model.y - eta) ** 2) / 2
                return 1 / (scale / host_rng.gamma(self.n_obs / 2, 1))
            out = np.empty(self.n_obs)
            shape = np.ascontiguousarray(model.n_trial, dtype=np.int32)
            _lib.check(self._lib.bbx_device_polya_gamma(
                dev, int(host_rng.integers(1, 2 ** 62)), self.n_obs,
                _ptr(shape), _ptr(np.ascontiguousarray(eta)), _ptr(out)))
            return out

        def update_global_scale(gscale, beta, bridge_exp, method='sample'):
            # only reached from initialize_chain with method='optimize'
            # (bayesbridge.py:318-320); per-iteration updates run on the device
            if beta.size == 0:
                return 1.
            if method is None:
                return gscale
            if method == 'optimize':
                gscale = self._monte_carlo_em_global_scale(beta, bridge_exp)
            else:
                hyper = prior.param['gscale_neg_power']
                phi = host_rng.gamma(hyper['shape'] + beta.size / bridge_exp) \
                    / (hyper['rate'] + np.sum(np.abs(beta) ** bridge_exp))
                gscale = 1 / phi ** (1 / bridge_exp)
            return max(gscale, self._lower_bd(bridge_exp))

        def update_local_scale(gscale, beta, bridge_exp):
            if bridge_exp == 2:
                return .5 * np.ones(beta.size)
            out = np.empty(beta.size)
            tilt = np.ascontiguousarray((beta / gscale) ** 2)
            if beta.size:
                _lib.check(self._lib.bbx_device_tilted_stable(
                    dev, int(host_rng.integers(1, 2 ** 62)), beta.size,
                    float(bridge_exp / 2), _ptr(tilt), _ptr(out)))
            lscale = np.sqrt(.5 / out)
            lscale[lscale == 0] = 10e-16
            lscale[np.isinf(lscale)] = 2.0 / gscale
            return lscale

        coef, obs_prec, lscale, gscale, init_used, optim_info = \
            self._initialize_chain(init, bridge_exp, update_local_scale,
                                   update_obs_precision, update_global_scale,
                                   sampler)
        chain.set_state(coef, obs_prec, lscale, gscale)
        return init_used, optim_info

    def _device_collect(self, chain, seed, kept, n_unconv, n_iter, n_burnin,
                        thin, params_to_save, init_used, optim_info):
        """Sample-major device output -> the reference's `samples` layout."""
        model = self.model
        samples, sampling_info = self._pre_allocate(
            n_iter - n_burnin, thin, params_to_save)
        if n_unconv > 0:
            warn("The conjugate gradient algorithm did not achieve the "
                 "requested tolerance level in %d iteration(s)." % n_unconv)
        # the library stores samples sample-major; users get the MCMC index
        # last (gibbs_util.py:126-127)
        for key in ('coef', 'local_scale'):
            if key in samples and samples[key].size:
                samples[key][:] = kept[key].T
        if 'obs_prec' in samples:
            samples['obs_prec'][:] = kept['obs_prec'][:, 0] \
                if model.name == 'linear' else kept['obs_prec'].T
        for key in ('global_scale', 'logp'):
            if key in samples:
                samples[key][:] = kept[key]
        sampling_info['n_cg_iter'][:] = kept['n_cg_iter']
        coef, obs_prec, lscale, gscale = chain.get_state()
        mean, square, n_avg = chain.get_summary()
        extra = {
            '_random_gen_state': {'kind': 'philox', 'seed': seed,
                                  'iteration': chain.iteration},
            '_reg_coef_sampler_state': {'mean': mean, 'square': square,
                                        'n_averaged': n_avg},
        }
        return (samples, sampling_info, (coef, obs_prec, lscale, gscale),
                init_used, optim_info, extra)
