"""Shared seeded inputs for the parity tests (no reference, no GPU needed)."""
import numpy as np
import scipy.sparse as sparse

from bayesbridge_amd import simulate


def mixed_design(n, p, binary_frac=.5, freq=.1, seed=0):
    """Mixed Gaussian/binary CSR design with the reference's distribution
    (tests/helper.py:8-16 uses simulate_design(n, p, binary_frac=.9))."""
    return simulate.simulate_design_csr(n, p, binary_frac=binary_frac,
                                        binary_pred_freq=freq, seed=seed)


def cg_inputs(n, P, n_unshrunk=1, seed=0, flat_intercept=True):
    """Plausible inputs of one CG draw: Omega ~ PG-like positives, prior
    precisions spanning several decades (bridge prior), a warm start."""
    rng = np.random.default_rng(seed)
    omega = rng.gamma(2., .15, n) + 1e-3
    lam = np.exp(rng.normal(0., 1.5, P))
    sd_prior = .05 * lam / np.sqrt(1 + (.05 * lam / 2.) ** 2)
    phi = 1. / sd_prior
    if flat_intercept and n_unshrunk > 0:
        phi[0] = 0.
    z = rng.normal(0., 3., P)
    x0 = rng.normal(0., .05, P)
    sd = .5 + rng.random(P)
    eta1 = rng.standard_normal(n)
    eta2 = rng.standard_normal(P)
    return dict(obs_prec=omega, prior_prec_sqrt=phi, z=z, coef_cg_init=x0,
                coef_scaled_sd=sd, randn_n=eta1, randn_P=eta2,
                n_unshrunk=n_unshrunk)


def config2_small_problem(golden_dir):
    """The scaled-down BASELINE config 2 problem of
    tests/golden/chain_logit_binary_20000x1000_summary.npz, regenerated without
    the reference (bayesbridge_amd.simulate replays the reference's RNG calls)
    and checked against the checksums stored in the fixture."""
    import os
    from bayesbridge_amd import simulate
    g = np.load(os.path.join(golden_dir,
                             'chain_logit_binary_20000x1000_summary.npz'))
    n, p = (int(v) for v in g['shape'])
    X = simulate.simulate_design_csr(n, p, binary_frac=1.,
                                     binary_pred_freq=float(g['freq']),
                                     seed=111)
    assert X.nnz == int(g['nnz'])
    assert np.array_equal(X.indptr[-4:], g['indptr_tail'])
    chk = (X.indices.astype(np.int64) * (np.arange(X.nnz) % 1009 + 1)).sum()
    assert chk == int(g['indices_checksum'])
    beta = simulate.demo_beta(p)
    n_success, n_trial = simulate.simulate_outcome(X, beta, 'logit', seed=1)
    assert n_success.sum() == g['n_success_sum']
    assert np.array_equal(n_success[:32], g['n_success_head'])
    assert np.array_equal(n_trial[:32], g['n_trial_head'])
    return g, X, (n_success, n_trial)
