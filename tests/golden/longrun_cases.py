"""Problems and statistics of the long-run distribution-parity test (shared by
tests/golden/make_longrun.py, which runs the imported reference in the build
container, and tests/test_hip_longrun.py, which runs the device-RNG chain on
the GPU).  Plain NumPy/SciPy: no reference, no GPU.

The chain under test is BayesBridge.gibbs(coef_sampler_type='cg') with the
default ('device') random streams -- bayesbridge.py:210-240 in the reference.
Its Philox streams cannot be compared draw by draw with the reference's
MT19937 / PCG64 streams, so the comparison is on ergodic averages with
Monte-Carlo standard errors from batch means on BOTH sides.
"""
import numpy as np
import scipy.sparse as sparse

BURNIN = 1000            # discarded on both sides
BATCH = 500              # iterations per batch mean (>> autocorrelation time)
REF_CHAINS = 4           # independent reference chains per case
REF_KEEP = 25000         # kept iterations per reference chain
DEV_KEEP = 30000         # kept iterations of the one device chain
Z_MAX = 4.5              # |z| bound for every compared statistic


def _binary_design(n, p, freq, rng):
    """Column frequencies 0.5 * Beta(.5, .5 (.5 / freq - 1)) as
    simulate_data.py:109-112, rows without replacement."""
    f = .5 * rng.beta(.5, .5 * (.5 / freq - 1.), p)
    f = np.maximum(f, 4. / n)
    cols, rows = [], []
    for j in range(p):
        k = int(np.ceil(n * f[j]))
        rows.append(rng.choice(n, k, replace=False))
        cols.append(np.full(k, j))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    X = sparse.coo_matrix((np.ones(rows.size), (rows, cols)),
                          shape=(n, p)).tocsr()
    X.sort_indices()
    return X


def _logit_outcome(X, beta, intercept, n_trial, rng):
    eta = intercept + X.dot(beta)
    prob = 1. / (1. + np.exp(-eta))
    return rng.binomial(n_trial.astype(np.int64), prob).astype(np.float64)


def make_case(name):
    """dict(family, X, outcome, prior_kw, init, env): X a SciPy CSR or a dense
    array (f64), outcome = y or (n_success, n_trial), prior_kw the arguments
    of RegressionCoefPrior, env the environment the DEVICE side sets before it
    builds the design (layout choice only)."""
    if name == 'logit_mixed_ntrial':
        # sparse CSR, 6 Gaussian + 54 binary columns (tests/helper.py:13 uses
        # binary_frac=.9), several trials on the last 600 rows
        rng = np.random.default_rng(2101)
        n, p = 2000, 60
        Xb = _binary_design(n, 54, .1, rng)
        Xd = rng.standard_normal((n, 6))
        X = sparse.hstack((sparse.csr_matrix(Xd), Xb)).tocsr()
        X.sort_indices()
        beta = np.zeros(p)
        beta[:3] = (.8, -.5, .3)
        beta[6:12] = (1.5, -1.5, 1., -1., .5, .5)
        n_trial = np.ones(n)
        n_trial[1400:] = 1 + rng.integers(1, 6, 600)
        n_success = _logit_outcome(X, beta, -.5, n_trial, rng)
        return dict(family='logit', X=X, outcome=(n_success, n_trial),
                    prior_kw=dict(bridge_exponent=.5,
                                  regularizing_slab_size=2.),
                    init={'global_scale': .1}, env={})
    if name == 'logit_binary_packed':
        # all-ones CSR, ~30 entries per row: the value-free tiled layout with
        # ids packed in groups of five (forced: the builder picks it by itself
        # only from 80 MB of ids on), bridge exponent 1/4 (the lambda kernel's
        # general-exponent path), proper prior on the global scale
        rng = np.random.default_rng(2102)
        n, p = 3000, 100
        X = _binary_design(n, p, .3, rng)
        beta = np.zeros(p)
        beta[:8] = (1.5, -1.5, 1., -1., .7, -.7, .4, .4)
        n_success = _logit_outcome(X, beta, .3, np.ones(n), rng)
        return dict(family='logit', X=X, outcome=(n_success, np.ones(n)),
                    prior_kw=dict(bridge_exponent=.25,
                                  regularizing_slab_size=1.5,
                                  global_scale_prior_hyper_param={
                                      'log10_mean': -2., 'log10_sd': 1.}),
                    init={'global_scale': .05},
                    env={'BBX_TILED_PACK': '1'})
    if name == 'linear_dense':
        # dense Gaussian design, linear model (obs_prec is the scalar noise
        # precision), bridge exponent 1 (Bayesian lasso; a = 1/2 in the
        # tilted-stable sampler)
        rng = np.random.default_rng(2103)
        n, p = 1500, 40
        X = rng.standard_normal((n, p))
        beta = np.zeros(p)
        beta[:6] = (1., -1., .5, -.5, .2, .1)
        y = 1.5 + X.dot(beta) + 2. * rng.standard_normal(n)
        return dict(family='linear', X=X, outcome=y,
                    prior_kw=dict(bridge_exponent=1.,
                                  regularizing_slab_size=2.),
                    init={'global_scale': .1}, env={})
    raise KeyError(name)


CASES = ('logit_mixed_ntrial', 'logit_binary_packed', 'linear_dense')


def case_checksum(case):
    """A few numbers that pin the regenerated problem to the one the fixture
    was made from."""
    X = case['X']
    if sparse.issparse(X):
        xs = np.array([X.nnz, X.indices.astype(np.int64).dot(
            np.arange(X.nnz) % 1009 + 1), X.data.sum()], dtype=np.float64)
    else:
        xs = np.array([X.size, X.sum(), np.abs(X).sum()])
    out = case['outcome']
    ys = np.array([np.sum(out[0]), np.sum(out[1])]) \
        if isinstance(out, tuple) else np.array([out.sum(), np.abs(out).sum()])
    return np.concatenate((xs, ys))


def series_names(case):
    X = case['X']
    p = X.shape[1]
    names = ['coef[%d]' % j for j in range(p + 1)]
    names += ['log_global_scale', 'logp']
    names += ['log_local_scale[%d]' % j for j in range(p)]
    if case['family'] == 'logit':
        names += ['mean_obs_prec', 'mean_obs_prec_single_trial',
                  'mean_obs_prec_multi_trial']
    else:
        names += ['log_obs_prec']
    return names


def series(case, samples):
    """Per-iteration statistics [T, K] of a `samples` dict in the reference's
    layout (MCMC index last; bayesbridge.py:162-167), saved with
    params_to_save='all'."""
    cols = [samples['coef'].T, np.log(samples['global_scale'])[:, None],
            samples['logp'][:, None], np.log(samples['local_scale']).T]
    if case['family'] == 'logit':
        om = samples['obs_prec']                 # [n, T]
        n_trial = case['outcome'][1]
        single = n_trial == 1
        cols.append(om.mean(axis=0)[:, None])
        cols.append(om[single].mean(axis=0)[:, None])
        cols.append((om[~single].mean(axis=0) if (~single).any()
                     else om.mean(axis=0))[:, None])
    else:
        cols.append(np.log(np.asarray(samples['obs_prec']).reshape(-1, 1)))
    return np.concatenate(cols, axis=1)


def batch_stats(S_list, batch=BATCH):
    """S_list: per chain an array [T, K] of per-iteration statistics.  Returns
    dict(mean, mean_se, var, var_se, n_batch): pooled ergodic mean and variance
    of every column with batch-means standard errors (batches never straddle
    chains)."""
    K = S_list[0].shape[1]
    total = sum(len(S) for S in S_list)
    mean = sum(S.sum(axis=0) for S in S_list) / total
    bm, bv = [], []
    for S in S_list:
        nb = len(S) // batch
        B = S[:nb * batch].reshape(nb, batch, K)
        bm.append(B.mean(axis=1))
        bv.append(((B - mean) ** 2).mean(axis=1))
    bm, bv = np.concatenate(bm), np.concatenate(bv)
    nb = len(bm)
    return dict(mean=bm.mean(axis=0), mean_se=bm.std(axis=0, ddof=1) / np.sqrt(nb),
                var=bv.mean(axis=0), var_se=bv.std(axis=0, ddof=1) / np.sqrt(nb),
                n_batch=nb)


def z_scores(a, b):
    """(z of the means, z of the variances) between two batch_stats dicts.
    The variance of side b is re-centred on side a's mean through
    E(x - m_a)^2 = var_b + (m_b - m_a)^2 only implicitly: both sides centre
    on their own mean, and a difference of the means is caught by the first
    z."""
    zm = (a['mean'] - b['mean']) / np.sqrt(a['mean_se'] ** 2
                                            + b['mean_se'] ** 2)
    zv = (a['var'] - b['var']) / np.sqrt(a['var_se'] ** 2 + b['var_se'] ** 2)
    return zm, zv
