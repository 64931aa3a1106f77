"""Generates tests/golden/longrun_<case>.npz: ergodic means / variances with
batch-means standard errors of REF_CHAINS x REF_KEEP post-burn-in iterations
of the IMPORTED REFERENCE (bayesbridge.py:109-277, coef_sampler_type='cg') on
the three problems of longrun_cases.py.  Build container only (ref_import.py);
the fixtures hold numbers only (a few KB each).

    python tests/golden/make_longrun.py [case ...]

One reference chain per process (the reference uses the process-global NumPy
stream, cg_sampler.py:61-62), REF_CHAINS processes side by side.
"""
import multiprocessing as mp
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import longrun_cases as lc  # noqa: E402


def run_reference_chain(args):
    name, seed, keep, burn = args
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
    import ref_import
    warnings.simplefilter('ignore')
    bb, _ = ref_import.import_reference()
    case = lc.make_case(name)
    X = case['X'].copy()          # DenseDesignMatrix centres in place
    model = bb.RegressionModel(case['outcome'], X, case['family'])
    prior = bb.RegressionCoefPrior(**case['prior_kw'])
    t0 = time.time()
    samples, info = bb.BayesBridge(model, prior).gibbs(
        burn + keep, n_burnin=burn, seed=seed, init=dict(case['init']),
        params_to_save='all', coef_sampler_type='cg')
    S = lc.series(case, samples)
    n_cg = info['_reg_coef_sampling_info']['n_cg_iter']
    return S, float(np.mean(n_cg)), time.time() - t0


def make(name, keep=lc.REF_KEEP, burn=lc.BURNIN, chains=lc.REF_CHAINS):
    case = lc.make_case(name)
    seeds = [1000 + 17 * k for k in range(chains)]
    with mp.get_context('spawn').Pool(chains) as pool:
        res = pool.map(run_reference_chain,
                       [(name, s, keep, burn) for s in seeds])
    S_list = [r[0] for r in res]
    pooled = lc.batch_stats(S_list)
    per_chain = [lc.batch_stats([S]) for S in S_list]
    # how far the reference's own chains sit from each other: the worst |z|
    # between one chain and the pool of the others (sanity of the errors)
    worst = 0.
    for k in range(chains):
        others = lc.batch_stats([S for i, S in enumerate(S_list) if i != k])
        zm, zv = lc.z_scores(per_chain[k], others)
        worst = max(worst, np.abs(zm).max(), np.abs(zv).max())
    names = lc.series_names(case)
    assert len(names) == S_list[0].shape[1]
    np.savez_compressed(
        os.path.join(HERE, 'longrun_%s.npz' % name),
        names=np.array(names), checksum=lc.case_checksum(case),
        mean=pooled['mean'], mean_se=pooled['mean_se'], var=pooled['var'],
        var_se=pooled['var_se'], n_batch=pooled['n_batch'],
        chain_mean=np.stack([c['mean'] for c in per_chain]),
        chain_var=np.stack([c['var'] for c in per_chain]),
        seeds=np.array(seeds), keep=keep, burnin=burn, batch=lc.BATCH,
        mean_n_cg=np.array([r[1] for r in res]),
        worst_z_between_reference_chains=worst)
    print('%s: %d chains x %d kept, %.0f s per chain, mean n_cg %.1f, worst '
          '|z| between reference chains %.2f'
          % (name, chains, keep, max(r[2] for r in res),
             np.mean([r[1] for r in res]), worst))


if __name__ == '__main__':
    for name in (sys.argv[1:] or lc.CASES):
        make(name)
