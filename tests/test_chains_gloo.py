"""CPU, world_size 2 over gloo: the multi-chain plumbing (per-rank seeds, the
single gather of samples at the end, max-over-ranks timing) that the 8-GPU
bench runs over RCCL.  The data path has no collective; this covers the only
exchange step."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.path.join(%(root)r, "bayes-bridge_amd"))
    os.environ["BBX_NO_TORCH"] = "0"
    import numpy as np
    import torch
    import torch.distributed as dist
    from bayesbridge_amd import chains
    rank, world, local_rank = chains.init_process_group_from_env(backend="gloo")
    assert world == 2 and dist.get_backend() == "gloo"
    seed = chains.chain_seed(111, rank)
    rng = np.random.default_rng(seed)
    local = torch.from_numpy(rng.standard_normal((5, 7)))      # [n_sample, P]
    chains.barrier()
    got = chains.gather_chain_samples(local, dst=0)
    t = chains.max_over_ranks(1.0 + rank)
    assert t == 2.0
    if rank == 0:
        assert got.shape == (2, 5, 7)
        for r in range(2):
            want = np.random.default_rng(111 + r).standard_normal((5, 7))
            assert np.array_equal(got[r].numpy(), want)
        merged = chains.merge_chain_outputs(got)
        assert merged['coef'].shape == (2, 7, 5)   # chain, coef, MCMC index last
        print("GATHER_OK")
    else:
        assert got is None
    assert chains.split_chains(5, 2, rank) == ([0, 2, 4] if rank == 0 else [1, 3])
    dist.destroy_process_group()
""")


def test_two_rank_gather_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
         "29533", str(script)],
        env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK" in out.stdout


def test_single_process_helpers_need_no_group():
    import torch
    from bayesbridge_amd import chains
    x = torch.arange(6, dtype=torch.float64).reshape(2, 3)
    got = chains.gather_chain_samples(x)
    assert got.shape == (1, 2, 3) and torch.equal(got[0], x)
    assert chains.max_over_ranks(3.5) == 3.5
    assert chains.chain_seed(111, 7) == 118      # BASELINE config 5 seeds
