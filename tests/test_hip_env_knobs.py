"""GPU: every SUPPORTED environment knob of libbbx.so once.
The knobs are read once per process, so every variant is its own short run of
scripts/chain_variant_run.py (a seeded device chain, samples to an .npz).

Scheduling knobs (streams, wave priorities, what is enqueued ahead, how many
elements a lane owns) must not change a single bit of the samples: Philox
streams are keyed by (seed, iteration, stream, element).  Knobs that choose
between kernels with differently ordered sums (the 3- / 4-launch CG iteration,
the dense block of a mixed design in the epilogue / in one pass / in separate
kernels) agree to rounding: the reference's CPU-vs-GPU bound, atol 1e-5
(tests/gpu_tests/test_gibbs.py:44), on the first draws and +-2 CG iterations."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

KEYS = ('coef', 'local_scale', 'obs_prec', 'global_scale', 'logp', 'n_cg_iter')
_cache = {}


def _run(tmp_path_factory, env, family='logit', n=4000, p=300, iters=4,
         binary_frac=.8, freq=.1):
    key = (tuple(sorted(env.items())), family, n, p, iters, binary_frac, freq)
    if key in _cache:
        return _cache[key]
    out = str(tmp_path_factory.mktemp("knob") / "chain.npz")
    full = dict(os.environ, BBX_NO_TORCH="1")
    full.update(env)
    run = subprocess.run(
        [sys.executable, os.path.join(ROOT, "scripts", "chain_variant_run.py"),
         out, family, str(n), str(p), str(iters), str(binary_frac), str(freq)],
        env=full, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    _cache[key] = (dict(np.load(out)), run.stdout)
    return _cache[key]


def _bitwise(a, b, what):
    for key in KEYS:
        assert np.array_equal(a[key], b[key]), (what, key)
    assert np.all(np.isfinite(a['logp']))


def _rounding(a, b, what):
    assert np.abs(a['n_cg_iter'] - b['n_cg_iter'])[:2].max() <= 2, what
    assert np.allclose(a['coef'][:2], b['coef'][:2], atol=1e-5), what
    assert np.allclose(a['global_scale'][:2], b['global_scale'][:2],
                       rtol=1e-4), what
    assert np.all(np.isfinite(a['logp'])) and np.all(np.isfinite(b['logp']))


# (knob, value, extra environment both runs share, design, comparison)
BINARY = dict(binary_frac=1.)           # all-ones tiled design, one column group
MIXED60 = dict()                        # 60 Gaussian columns: one-pass dense block
MIXED6 = dict(p=30)                     # 6 Gaussian columns: dense block in the epilogue
FORK = {'BBX_CHAIN_FORK': '1'}
CASES = [
    ('BBX_CHAIN_TAIL', '0', {}, BINARY, 'bitwise'),
    ('BBX_CHAIN_FORK', '1', {}, BINARY, 'bitwise'),
    ('BBX_ETA_AHEAD', '1', FORK, BINARY, 'bitwise'),
    ('BBX_FILL_PRIO', '0', dict(FORK, BBX_ETA_AHEAD='1'), BINARY, 'bitwise'),
    ('BBX_BRANCH_PRIO', '0', {}, BINARY, 'bitwise'),
    ('BBX_LSCALE_PRIO', '0', {}, BINARY, 'bitwise'),
    ('BBX_TS_ITEMS', '64', {}, BINARY, 'bitwise'),
    ('BBX_PG_ELEMS', '1', {}, BINARY, 'bitwise'),
    ('BBX_PG_ELEMS', '8', {}, BINARY, 'bitwise'),
    ('BBX_TILED_STATS', '1', {}, BINARY, 'bitwise'),
    ('BBX_BUILD_THREADS', '1', {}, BINARY, 'bitwise'),
    ('BBX_CG_AHEAD', '1', {}, BINARY, 'bitwise'),
    ('BBX_CG_AHEAD', '7', {}, BINARY, 'bitwise'),
    ('BBX_CG_SLEEP', '0', {'BBX_CG_AHEAD': '1'}, BINARY, 'bitwise'),
    ('BBX_CG_FOLD', '0', {}, BINARY, 'rounding'),
    ('BBX_TILED_PR', '1024', {}, BINARY, 'rounding'),
    ('BBX_TILED_PACK', '1', {}, BINARY, 'rounding'),
    ('BBX_HYB_FUSED', '0', {}, MIXED60, 'rounding'),
    ('BBX_DENSE_EPI_MAX', '0', {}, MIXED6, 'rounding'),
]


@pytest.mark.parametrize("knob,value,shared,design,how", CASES,
                         ids=["%s=%s" % c[:2] for c in CASES])
def test_supported_knob(tmp_path_factory, knob, value, shared, design, how):
    base, _ = _run(tmp_path_factory, dict(shared), **design)
    var, log = _run(tmp_path_factory, dict(shared, **{knob: value}), **design)
    (_bitwise if how == 'bitwise' else _rounding)(var, base, knob)
    if knob == 'BBX_CG_FOLD':
        # the knob did switch the loop: 3 launches per CG iteration by
        # default at this size, 4 with the fold off
        assert "cg launches 4" in log
        assert "cg launches 3" in _run(tmp_path_factory, {}, **design)[1]
    assert base['coef'].shape[0] == 4


def test_one_lane_polya_gamma_kernel_of_rounds_1_to_4_still_runs(
        tmp_path_factory):
    """BBX_PG_ELEMS=0 (diagnostic: the sequential sampler, one lane per draw)
    draws from the same distribution through another consumption of the
    streams: finite samples, same CG effort."""
    base, _ = _run(tmp_path_factory, {}, **BINARY)
    old, _ = _run(tmp_path_factory, {'BBX_PG_ELEMS': '0'}, **BINARY)
    assert np.all(np.isfinite(old['logp'])) and np.all(old['obs_prec'] > 0)
    assert abs(old['n_cg_iter'].mean() - base['n_cg_iter'].mean()) < 6
    assert abs(old['obs_prec'].mean() / base['obs_prec'].mean() - 1) < .02


def test_host_sleeping_between_stop_tests_changes_no_bit(tmp_path_factory):
    """BBX_CG_SLEEP=1 (what a rank does by itself with fewer than three host
    cores): on a design whose CG iteration outlasts 60 us -- 1M x 2000, ~8e7
    entries -- the host sleeps through three quarters of every iteration
    instead of polling the progress word.  Same samples bit for bit, same
    stopping iterations; three launches enqueued in vain per draw either way."""
    big = dict(n=1000000, p=2000, iters=3, binary_frac=1., freq=.04)
    # (the design is generated in HBM with torch: no BBX_NO_TORCH here)
    poll, log_p = _run(tmp_path_factory, {'BBX_CG_SLEEP': '0',
                                          'BBX_NO_TORCH': '0'}, **big)
    nap, log_n = _run(tmp_path_factory, {'BBX_CG_SLEEP': '1',
                                         'BBX_NO_TORCH': '0'}, **big)
    _bitwise(nap, poll, 'BBX_CG_SLEEP')
    assert "cg launches 4" in log_p and "format tiled" in log_p
    # the path under test was taken: several naps per solve, none when polling
    assert "naps 0" in log_p
    naps = int(log_n.split("naps")[1].split()[0])
    assert naps >= 3 * 5, log_n
    assert poll['n_cg_iter'].min() >= 5
