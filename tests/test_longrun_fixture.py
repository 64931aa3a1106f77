"""CPU: the long-run fixtures (tests/golden/longrun_*.npz, made by
make_longrun.py from 4 x 25 000 iterations of the imported reference) belong
to the problems longrun_cases.py regenerates, and their Monte-Carlo standard
errors are sane: every one of the four reference chains sits within 5
standard errors of the pool -- the yardstick the GPU test then applies to the
device chain."""
import os

import numpy as np
import pytest

import longrun_cases as lc


@pytest.mark.parametrize("name", lc.CASES)
def test_fixture_matches_problem_and_reference_chains_agree(golden_dir, name):
    g = np.load(os.path.join(golden_dir, 'longrun_%s.npz' % name))
    case = lc.make_case(name)
    assert np.allclose(lc.case_checksum(case), g['checksum'], rtol=1e-12)
    assert list(g['names']) == lc.series_names(case)
    assert int(g['keep']) == lc.REF_KEEP and int(g['burnin']) == lc.BURNIN
    n_chain = len(g['seeds'])
    assert n_chain == lc.REF_CHAINS
    assert int(g['n_batch']) == n_chain * (lc.REF_KEEP // lc.BATCH)
    assert np.all(g['mean_se'] > 0) and np.all(g['var_se'] > 0)
    assert np.all(g['var'] > 0)
    # chain k against the pool that contains it: the variance of the
    # difference is se_pool^2 (n_chain - 1)
    z = (g['chain_mean'] - g['mean']) / (g['mean_se'] * np.sqrt(n_chain - 1))
    assert np.abs(z).max() < 5., np.abs(z).max()
    assert float(g['worst_z_between_reference_chains']) < 4.5
    # the relative error of an ergodic mean is small where the test has
    # power: the signal coefficients are known to better than 1 % of their
    # posterior sd x 10
    sd = np.sqrt(g['var'])
    assert np.median(g['mean_se'] / sd) < .02


def test_batch_statistics_on_white_noise():
    """batch_stats / z_scores on i.i.d. draws: unbiased means and variances,
    standard errors that cover."""
    rng = np.random.default_rng(1)
    a = [rng.standard_normal((10000, 30)) * 2. + 1. for _ in range(2)]
    b = [rng.standard_normal((15000, 30)) * 2. + 1.]
    sa, sb = lc.batch_stats(a), lc.batch_stats(b)
    assert sa['n_batch'] == 40 and sb['n_batch'] == 30
    assert np.abs(sa['mean'] - 1.).max() < 5 * 2. / np.sqrt(20000)
    assert np.abs(sa['var'] - 4.).max() < 5 * 4. * np.sqrt(2. / 20000)
    assert np.allclose(sa['mean_se'], 2. / np.sqrt(20000), rtol=.5)
    zm, zv = lc.z_scores(sa, sb)
    assert np.abs(zm).max() < 4.5 and np.abs(zv).max() < 4.5
    # a 5 % scale error is visible in the variances of 30 columns
    c = lc.batch_stats([rng.standard_normal((15000, 30)) * 2.1 + 1.])
    _, zv = lc.z_scores(c, sa)
    assert np.sqrt((zv ** 2).mean()) > 3.
