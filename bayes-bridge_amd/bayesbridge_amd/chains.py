"""Independent Gibbs chains, one per GPU (one process per GPU), with a single
gather of the samples at the end.

The reference has no multi-chain API (one chain per process, no communication;
bayesbridge.py:109).  Chains are embarrassingly parallel: every rank holds a
full replica of X and its own seed, nothing is exchanged while sampling, and
the kept samples are gathered once over RCCL/xGMI (`backend='nccl'` IS RCCL on
ROCm; `gloo` on CPU for the tests).  SURVEY.md 8(e).
"""
import os

import numpy as np


def init_process_group_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)
    and returns (rank, world_size, local_rank).  World size 1 needs no group."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # RCCL needs one distinct GPU per rank; with fewer devices than
            # ranks (dry runs on a 1-GPU box) fall back to gloo
            backend = "nccl" if (torch.cuda.is_available() and
                                 torch.cuda.device_count() >= world) else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def chain_seed(base_seed, rank):
    """Chain k uses seed base + k (BASELINE config 5: seeds 111...118)."""
    return int(base_seed) + int(rank)


def gather_chain_samples(local, dst=0):
    """Gathers one tensor per rank (same shape everywhere) on `dst`.
    Returns a tensor [world, *local.shape] on dst, None elsewhere.  A gather
    to one root uses the root's inbound xGMI links concurrently, which suits
    the point-to-point fabric better than a ring."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) \
            or dist.get_world_size() == 1:
        return local.unsqueeze(0)
    world = dist.get_world_size()
    rank = dist.get_rank()
    local = local.contiguous()
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()
    if rank == dst:
        bucket = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=bucket, dst=dst)
        return torch.stack(bucket)
    dist.gather(local, gather_list=None, dst=dst)
    return None


def max_over_ranks(value):
    """MAX all-reduce of a Python float (timing: the slowest rank counts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) \
            or dist.get_world_size() == 1:
        return float(value)
    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def split_chains(n_chain, world, rank):
    """Chains handled by `rank` when there are more chains than ranks."""
    return list(range(rank, n_chain, world))


def merge_chain_outputs(gathered, params=('coef',)):
    """[world, n_sample, P] -> dict of NumPy arrays with the chain index first
    and, as in the reference's `samples`, the MCMC index last."""
    arr = gathered.cpu().numpy()
    return {'coef': np.transpose(arr, (0, 2, 1))}
