"""Thin object wrapper of the `bbx_chain_*` entry points (include/bbx.h): one
device-resident Gibbs chain of BayesBridge.gibbs(coef_sampler_type='cg')
(bayesbridge.py:210-240) bound to one HipDesignMatrix.

`BayesBridge._gibbs_device`, `bench.py` and the parity tests all drive the
chain through this class; it adds nothing to the C ABI but NumPy marshalling.
"""
from ctypes import byref, c_double, c_int64, c_uint64, c_void_p

import numpy as np

from . import _lib

_GSCALE_MODES = {'sample': _lib.GSCALE_SAMPLE, 'optimize': _lib.GSCALE_OPTIMIZE,
                 None: _lib.GSCALE_FIXED}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class HipGibbsChain():

    def __init__(self, design, family, outcome, n_trial=None,
                 sd_unshrunk=(float('inf'),), bridge_exponent=.5,
                 slab_size=float('inf'), gscale_shape=0., gscale_rate=0.,
                 seed=0):
        """family: 'linear' (outcome = y) or 'logit' (outcome = n_success);
        sd_unshrunk: prior sd of the unshrunk coefficients, intercept first
        (bayesbridge.py:26-32); gscale_shape/rate: Gamma prior on
        tau^-bridge_exponent (prior.py:77-81), raw parametrisation."""
        if family not in ('linear', 'logit'):
            raise ValueError("family must be 'linear' or 'logit'")
        self._lib = _lib.load()
        self._c = c_void_p()
        self.design = design            # keeps the operator alive
        self.family = family
        self.n, self.P = design.shape
        sd = _f64(np.atleast_1d(sd_unshrunk))
        self.n_unshrunk = int(sd.size)
        outcome = _f64(outcome)
        if outcome.shape != (self.n,):
            raise ValueError("outcome must have one entry per design row")
        if n_trial is not None:
            n_trial = _f64(n_trial)
        _lib.check(self._lib.bbx_chain_create(
            design.handle,
            _lib.MODEL_LINEAR if family == 'linear' else _lib.MODEL_LOGIT,
            _ptr(outcome), _ptr(n_trial), self.n_unshrunk, _ptr(sd),
            float(bridge_exponent), float(slab_size), float(gscale_shape),
            float(gscale_rate), int(seed) & 0xFFFFFFFFFFFFFFFF,
            byref(self._c)))

    def close(self):
        c = getattr(self, '_c', None)
        if c is not None and c.value and not getattr(_lib, "finalizing", True):
            try:
                self._lib.bbx_chain_destroy(c)
            except Exception:
                pass
            self._c = c_void_p()

    __del__ = close

    @property
    def handle(self):
        return self._c

    @property
    def n_shrunk(self):
        return self.P - self.n_unshrunk

    @property
    def _obs_len(self):
        return self.n if self.family == 'logit' else 1

    # ---- Markov-chain state (raw parametrisation, prior.py:129-141)
    def set_state(self, coef=None, obs_prec=None, local_scale=None,
                  global_scale=None):
        coef = None if coef is None else _f64(coef)
        obs = None if obs_prec is None else _f64(np.atleast_1d(obs_prec))
        ls = None if local_scale is None else _f64(local_scale)
        if coef is not None and coef.shape != (self.P,):
            raise ValueError("coef must have length %d" % self.P)
        if obs is not None and obs.shape != (self._obs_len,):
            raise ValueError("obs_prec must have length %d" % self._obs_len)
        if ls is not None and ls.shape != (self.n_shrunk,):
            raise ValueError("local_scale must have length %d" % self.n_shrunk)
        g = None if global_scale is None else c_double(float(global_scale))
        _lib.check(self._lib.bbx_chain_set_state(
            self._c, _ptr(coef), _ptr(obs), _ptr(ls),
            None if g is None else byref(g)))

    def get_state(self):
        """(coef, obs_prec, local_scale, global_scale); obs_prec is a float
        for the linear model."""
        coef = np.empty(self.P)
        ls = np.empty(self.n_shrunk)
        obs = np.empty(self._obs_len)
        g = c_double()
        _lib.check(self._lib.bbx_chain_get_state(
            self._c, _ptr(coef), _ptr(obs), _ptr(ls), byref(g)))
        return (coef, obs if self.family == 'logit' else float(obs[0]), ls,
                float(g.value))

    def init_obs_prec(self):
        _lib.check(self._lib.bbx_chain_init_obs_prec(self._c))

    # ---- running summaries (reg_coef_posterior_summarizer.py:68-124)
    def set_summary(self, mean, square, n_averaged):
        mean, square = _f64(mean), _f64(square)
        if mean.shape != (self.P,) or square.shape != (self.P,):
            raise ValueError("summaries must have length %d" % self.P)
        _lib.check(self._lib.bbx_chain_set_summary(
            self._c, _ptr(mean), _ptr(square), int(n_averaged)))

    def get_summary(self):
        mean, square, n_avg = np.empty(self.P), np.empty(self.P), c_int64()
        _lib.check(self._lib.bbx_chain_get_summary(
            self._c, _ptr(mean), _ptr(square), byref(n_avg)))
        return mean, square, int(n_avg.value)

    # ---- Philox bookkeeping
    @property
    def iteration(self):
        it = c_int64()
        _lib.check(self._lib.bbx_chain_get_iteration(self._c, byref(it)))
        return int(it.value)

    @iteration.setter
    def iteration(self, value):
        _lib.check(self._lib.bbx_chain_set_iteration(self._c, int(value)))

    @property
    def seed(self):
        s = c_uint64()
        _lib.check(self._lib.bbx_chain_get_seed(self._c, byref(s)))
        return int(s.value)

    @seed.setter
    def seed(self, value):
        _lib.check(self._lib.bbx_chain_set_seed(
            self._c, int(value) & 0xFFFFFFFFFFFFFFFF))

    def set_gscale_update(self, method):
        """'sample' | 'optimize' | None (SamplerOptions.gscale_update)."""
        _lib.check(self._lib.bbx_chain_set_gscale_update(
            self._c, _GSCALE_MODES[method]))

    def eta(self, iteration):
        """(eta1[n], eta2[P]): the normals of the CG draw at 0-based
        `iteration` (cg_sampler.py:61-62), regenerated from the counters."""
        e1, e2 = np.empty(self.n), np.empty(self.P)
        _lib.check(self._lib.bbx_chain_eta(self._c, int(iteration), _ptr(e1),
                                           _ptr(e2)))
        return e1, e2

    def set_progress(self, every, fn=None):
        """fn(iteration) is called every `every` iterations of run() /
        run_device() (BayesBridge.gibbs(n_status_update=...)); every = 0 or
        fn = None switches it off."""
        import ctypes
        if fn is None or not every:
            self._progress_cb = None
            _lib.check(self._lib.bbx_chain_set_progress(self._c, 0, None, None))
            return
        proto = ctypes.CFUNCTYPE(None, ctypes.c_int, c_void_p)
        self._progress_cb = proto(lambda it, _ctx: fn(int(it)))   # kept alive
        _lib.check(self._lib.bbx_chain_set_progress(
            self._c, int(every), ctypes.cast(self._progress_cb, c_void_p),
            None))

    def logp(self):
        """(log-likelihood, log-posterior) left by the last iteration."""
        ll, lp = c_double(), c_double()
        _lib.check(self._lib.bbx_chain_get_logp(self._c, byref(ll), byref(lp)))
        return float(ll.value), float(lp.value)

    # ---- sampling
    def run(self, n_iter, n_burnin=0, thin=1, maxiter=500, atol=0.,
            save=('coef',)):
        """Runs n_iter Gibbs iterations on the device and returns
        (samples, n_unconverged).  samples: sample-major arrays for the names
        in `save` (out of 'coef', 'local_scale', 'obs_prec') plus
        'global_scale', 'logp', 'n_cg_iter' (always)."""
        n_sample = (n_iter - n_burnin) // thin
        rows = max(n_sample, 1)
        bufs = {
            'coef': np.zeros((rows, self.P)) if 'coef' in save else None,
            'local_scale': np.zeros((rows, max(self.n_shrunk, 1)))
            if 'local_scale' in save else None,
            'obs_prec': np.zeros((rows, self._obs_len))
            if 'obs_prec' in save else None,
        }
        gs, lp, ncg = np.zeros(rows), np.zeros(rows), np.zeros(rows)
        n_unconv = _lib.check(self._lib.bbx_chain_run_host(
            self._c, int(n_iter), int(n_burnin), int(thin), int(maxiter),
            float(atol), _ptr(bufs['coef']), _ptr(bufs['local_scale']),
            _ptr(bufs['obs_prec']), _ptr(gs), _ptr(lp), _ptr(ncg)))
        out = {'global_scale': gs[:n_sample], 'logp': lp[:n_sample],
               'n_cg_iter': ncg[:n_sample]}
        if bufs['coef'] is not None:
            out['coef'] = bufs['coef'][:n_sample]
        if bufs['local_scale'] is not None:
            out['local_scale'] = bufs['local_scale'][:n_sample,
                                                     :self.n_shrunk]
        if bufs['obs_prec'] is not None:
            out['obs_prec'] = bufs['obs_prec'][:n_sample]
        return out, n_unconv

    def run_device(self, n_iter, d_coef_ptr=None, n_burnin=0, thin=1,
                   maxiter=500, atol=0., d_lscale_ptr=None,
                   d_obs_prec_ptr=None):
        """Same, but the kept samples go to DEVICE buffers (raw pointers,
        sample-major: coef [n_sample, P], local_scale [n_sample, P -
        n_unshrunk], obs_prec [n_sample, n] (logit) or [n_sample] (linear));
        nothing but the per-sample scalars crosses PCIe.  Returns
        (global_scale, logp, n_cg_iter, n_unconverged)."""
        n_sample = (n_iter - n_burnin) // thin
        rows = max(n_sample, 1)
        gs, lp, ncg = np.zeros(rows), np.zeros(rows), np.zeros(rows)

        def dev(p):
            return c_void_p(int(p)) if p else None
        n_unconv = _lib.check(self._lib.bbx_chain_run(
            self._c, int(n_iter), int(n_burnin), int(thin), int(maxiter),
            float(atol), dev(d_coef_ptr), dev(d_lscale_ptr),
            dev(d_obs_prec_ptr), _ptr(gs), _ptr(lp), _ptr(ncg)))
        return gs[:n_sample], lp[:n_sample], ncg[:n_sample], n_unconv


class HipChainBatch():
    """Chains that share every pass over the design (`bbx_batch_*`): the
    products of the CG solves and the linear predictor of the Omega update run
    once for all chains of the batch (K-column products over one read of the
    matrix).  The reference has one chain per process (bayesbridge.py:109);
    this is how more chains than GPUs are run here.  `chains`: 2 or 4
    HipGibbsChain objects on the same design (2 ... 32 on dense designs); set
    / get their state through the chains themselves between runs.

    A width the library's cost model prices below single chains
    (`predicted_speedup(design, k) < 1`: four chains on a 1M x 50k sparse
    design, two on an f32 dense one) raises; `allow_slow=True` builds it
    anyway (parity tests, measurements)."""

    def __init__(self, chains, allow_slow=False):
        import ctypes
        chains = list(chains)
        self._lib = _lib.load()
        self._b = c_void_p()
        self.chains = chains             # keeps them (and the design) alive
        self.design = chains[0].design
        self.P = chains[0].P
        arr = (c_void_p * len(chains))(*[c.handle.value for c in chains])
        _lib.check(self._lib.bbx_batch_create_opts(
            self.design.handle, len(chains), ctypes.cast(arr, c_void_p),
            _lib.BATCH_ALLOW_SLOW if allow_slow else 0, byref(self._b)))

    @staticmethod
    def predicted_speedup(design, n_chain):
        """The cost model's estimate of (throughput of a batch of n_chain
        chains) / (the same chains one at a time) on `design`; no layout is
        built (`bbx_batch_predict`)."""
        s = c_double()
        _lib.check(_lib.load().bbx_batch_predict(design.handle, int(n_chain),
                                                 byref(s)))
        return float(s.value)

    @property
    def n_unconverged(self):
        """Per chain: CG solves of the last run() that hit maxiter."""
        from ctypes import c_int
        arr = (c_int * len(self.chains))()
        _lib.check(self._lib.bbx_batch_unconverged(self._b, arr))
        return [int(v) for v in arr]

    def close(self):
        b = getattr(self, '_b', None)
        if b is not None and b.value and not getattr(_lib, "finalizing", True):
            try:
                self._lib.bbx_batch_destroy(b)
            except Exception:
                pass
            self._b = c_void_p()

    __del__ = close

    @property
    def n_chain(self):
        return len(self.chains)

    @property
    def launch_bytes(self):
        """(dot, Tdot): bytes one batched launch of each product moves."""
        d, t = c_int64(), c_int64()
        _lib.check(self._lib.bbx_batch_bytes(self._b, byref(d), byref(t)))
        return int(d.value), int(t.value)

    def dot(self, v):
        """[n_chain, P] -> [n_chain, n]: X~ v_c through the batched kernel."""
        v = _f64(v)
        out = np.empty((self.n_chain, self.chains[0].n))
        _lib.check(self._lib.bbx_batch_dot(self._b, _ptr(v), _ptr(out)))
        return out

    def Tdot(self, w):
        """[n_chain, n] -> [n_chain, P]: X~^T w_c through the batched kernel."""
        w = _f64(w)
        out = np.empty((self.n_chain, self.P))
        _lib.check(self._lib.bbx_batch_tdot(self._b, _ptr(w), _ptr(out)))
        return out

    def run(self, n_iter, n_burnin=0, thin=1, maxiter=500, atol=0.,
            save_coef=True):
        """n_iter Gibbs iterations of every chain.  Returns (samples,
        n_unconverged); samples: 'coef' [n_chain, n_sample, P] (if
        save_coef), 'global_scale', 'logp', 'n_cg_iter' [n_chain, n_sample]."""
        K = self.n_chain
        n_sample = (n_iter - n_burnin) // thin
        rows = max(n_sample, 1)
        coef = np.zeros((K, rows, self.P)) if save_coef else None
        gs, lp, ncg = (np.zeros((K, rows)) for _ in range(3))
        n_unconv = _lib.check(self._lib.bbx_batch_run_host(
            self._b, int(n_iter), int(n_burnin), int(thin), int(maxiter),
            float(atol), _ptr(coef), _ptr(gs), _ptr(lp), _ptr(ncg)))
        out = {'global_scale': gs[:, :n_sample], 'logp': lp[:, :n_sample],
               'n_cg_iter': ncg[:, :n_sample]}
        if coef is not None:
            out['coef'] = coef[:, :n_sample]
        return out, n_unconv

    def run_device(self, n_iter, d_coef_ptrs=None, n_burnin=0, thin=1,
                   maxiter=500, atol=0.):
        """Same, the kept coefficients going to one DEVICE buffer per chain
        (raw pointers, each sample-major [n_sample, P]).  Returns
        (global_scale, logp, n_cg_iter, n_unconverged), arrays
        [n_chain, n_sample]."""
        import ctypes
        K = self.n_chain
        n_sample = (n_iter - n_burnin) // thin
        rows = max(n_sample, 1)
        gs, lp, ncg = (np.zeros((K, rows)) for _ in range(3))
        arr = None
        if d_coef_ptrs is not None:
            arr = (c_void_p * K)(*[int(p) for p in d_coef_ptrs])
        n_unconv = _lib.check(self._lib.bbx_batch_run(
            self._b, int(n_iter), int(n_burnin), int(thin), int(maxiter),
            float(atol), None if arr is None else ctypes.cast(arr, c_void_p),
            _ptr(gs), _ptr(lp), _ptr(ncg)))
        return gs[:, :n_sample], lp[:, :n_sample], ncg[:, :n_sample], n_unconv
