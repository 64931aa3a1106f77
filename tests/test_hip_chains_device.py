"""GPU: BayesBridge.gibbs_multichain / chains.run_chains hand the vector
samples to the collective straight from the device buffers bbx_chain_run
fills (no device -> NumPy -> device round trip before the gather), and what
arrives is bit for bit what bridge.gibbs returns chain by chain.  The reference
has one chain per process (bayesbridge.py:109); SURVEY.md 8(e)."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bridge(family):
    from bayesbridge_amd import (BayesBridge, RegressionCoefPrior,
                                 RegressionModel, simulate)
    rng = np.random.default_rng(3)
    if family == 'logit':
        X = simulate.simulate_binary_csr_fast(3000, 120, .05, seed=5)
        beta = simulate.demo_beta(120)
        outcome = simulate.simulate_outcome(X, beta, 'logit', seed=1)
    else:
        X = rng.standard_normal((1200, 50))
        outcome = X[:, :3].sum(axis=1) + rng.standard_normal(1200)
    return BayesBridge(RegressionModel(outcome, X, family),
                       RegressionCoefPrior(bridge_exponent=.5,
                                           regularizing_slab_size=2.))


@pytest.mark.parametrize("family", ['logit', 'linear'])
def test_run_chains_device_slabs_equal_gibbs_chain_by_chain(family):
    bridge = _bridge(family)
    kw = dict(n_burnin=2, thin=2, init={'global_scale': .05})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        merged, infos = bridge.gibbs_multichain(
            3, 12, seed=40, params_to_save='all', **kw)
        singles = [bridge.gibbs(12, seed=40 + k, params_to_save='all',
                                coef_sampler_type='cg', **kw)
                   for k in range(3)]
    assert [i['chain'] for i in infos] == [0, 1, 2]
    for k, (s, info) in enumerate(singles):
        for name in ('coef', 'local_scale', 'obs_prec', 'global_scale',
                     'logp'):
            assert merged[name][k].shape == s[name].shape, name
            assert np.array_equal(merged[name][k], s[name]), (name, k)
        assert np.array_equal(
            merged['n_cg_iter'][k],
            info['_reg_coef_sampling_info']['n_cg_iter'])
    assert infos[0]['saved_params'] == ('coef', 'local_scale', 'global_scale',
                                        'logp', 'obs_prec')


def test_device_out_argument_is_checked():
    import torch
    bridge = _bridge('linear')
    P = bridge.n_pred
    good = torch.zeros((5, P), dtype=torch.float64, device='cuda')
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s, _ = bridge.gibbs(5, seed=1, _device_out={'coef': good})
        ref, _ = bridge.gibbs(5, seed=1)
    assert 'coef' not in s
    assert np.array_equal(good.cpu().numpy().T, ref['coef'])
    for bad in (torch.zeros((5, P), dtype=torch.float32, device='cuda'),
                torch.zeros((4, P), dtype=torch.float64, device='cuda'),
                torch.zeros((5, P), dtype=torch.float64)):
        with pytest.raises(ValueError):
            bridge.gibbs(5, seed=1, _device_out={'coef': bad})
    with pytest.raises(ValueError):
        bridge.gibbs(5, seed=1, options={'rng': 'reference'},
                     _device_out={'coef': good})


RCCL_WORKER = """
import os, sys, warnings
sys.path.insert(0, %(pkg)r)
sys.path.insert(0, %(tests)r)
import numpy as np
import torch
import torch.distributed as dist
from bayesbridge_amd import chains
import test_hip_chains_device as T
rank, world, _ = chains.init_process_group_from_env(
    backend='nccl', single_rank_group=True)
assert dist.get_backend() == 'nccl' and world == 1
seen = []
orig = chains.gather_chain_samples
def spy(local, dst=0):
    seen.append((tuple(local.shape), bool(local.is_cuda)))
    return orig(local, dst)
chains.gather_chain_samples = spy
bridge = T._bridge('logit')
warnings.simplefilter('ignore')
merged, infos = bridge.gibbs_multichain(2, 8, n_burnin=2, seed=7)
s0, _ = bridge.gibbs(8, n_burnin=2, seed=7)
s1, _ = bridge.gibbs(8, n_burnin=2, seed=8)
assert np.array_equal(merged['coef'][0], s0['coef'])
assert np.array_equal(merged['coef'][1], s1['coef'])
# the coefficient slab went into the RCCL gather as a DEVICE tensor
assert ((2, 6, bridge.n_pred), True) in seen, seen
dist.destroy_process_group()
print('DEVICE_GATHER_OK')
"""


def test_multichain_gathers_device_slabs_over_rccl_with_one_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER % {
        "pkg": os.path.join(ROOT, "bayes-bridge_amd"),
        "tests": os.path.join(ROOT, "tests")})
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29647",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "DEVICE_GATHER_OK" in out.stdout


TWO_RANK_WORKER = """
import os, sys, warnings
sys.path.insert(0, %(pkg)r)
sys.path.insert(0, %(tests)r)
import numpy as np
import torch
import torch.distributed as dist
from bayesbridge_amd import chains
import test_hip_chains_device as T
warnings.simplefilter('ignore')
rank, world, _ = chains.init_process_group_from_env()
assert world == 2
with chains.setup_turn():
    bridge = T._bridge('logit')
# five chains over two ranks: rank 0 runs chains 0, 2, 4, rank 1 chains 1, 3
merged, infos = bridge.gibbs_multichain(
    5, 10, n_burnin=2, thin=2, seed=300,
    params_to_save=('coef', 'local_scale', 'global_scale', 'logp'))
assert [i['chain'] for i in infos] == chains.split_chains(5, 2, rank)
if rank == 0:
    assert merged['coef'].shape == (5, bridge.n_pred, 4)
    for k in range(5):
        s, info = bridge.gibbs(
            10, n_burnin=2, thin=2, seed=300 + k,
            params_to_save=('coef', 'local_scale', 'global_scale', 'logp'))
        for name in ('coef', 'local_scale', 'global_scale', 'logp'):
            assert np.array_equal(merged[name][k], s[name]), (name, k)
        assert np.array_equal(merged['n_cg_iter'][k],
                              info['_reg_coef_sampling_info']['n_cg_iter'])
    print('TWO_RANK_OK backend=%%s' %% dist.get_backend())
else:
    assert merged is None
dist.destroy_process_group()
"""


def test_multichain_over_two_ranks_equals_single_chains(tmp_path):
    """config 5's API path in miniature on the GPU box: two ranks (sharing
    the one device, gloo; one rank per GPU over RCCL on a node) split five
    chains, every chain's device slab goes into the gather, and rank 0 holds
    bit for bit what bridge.gibbs returns seed by seed."""
    from bayesbridge_amd import chains
    script = tmp_path / "worker.py"
    script.write_text(TWO_RANK_WORKER % {
        "pkg": os.path.join(ROOT, "bayes-bridge_amd"),
        "tests": os.path.join(ROOT, "tests")})
    out = chains.launch_ranks(2, [str(script)], capture=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "TWO_RANK_OK" in out.stdout
