"""Independent Gibbs chains, one per GPU (one process per GPU), with a single
gather of the samples at the end.

The reference has no multi-chain API (one chain per process, no communication;
bayesbridge.py:109).  Chains are embarrassingly parallel: every rank holds a
full replica of X and its own seed, nothing is exchanged while sampling, and
the kept samples are gathered once over RCCL/xGMI (`backend='nccl'` IS RCCL on
ROCm; `gloo` on CPU for the tests).  SURVEY.md 8(e).
"""
import os
import socket
import subprocess
import sys

import numpy as np


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_ranks(n_ranks, script_argv, port=None, extra_env=None,
                 capture=False, timeout=None):
    """Starts `n_ranks` processes of `script_argv` (a Python script + its
    arguments) on this node through torch.distributed.run, one rank per GPU,
    as a CHILD process (never exec: a process that has touched the GPU must
    not be replaced) and returns its exit code -- or the CompletedProcess when
    `capture`.  Call it before anything in the parent initialises HIP."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % int(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port or free_port())] + list(script_argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL
    env.setdefault("OMP_NUM_THREADS", "1")
    if extra_env:
        env.update(extra_env)
    if capture:
        return subprocess.run(cmd, env=env, capture_output=True, text=True,
                              timeout=timeout)
    return subprocess.call(cmd, env=env, timeout=timeout)


def init_process_group_from_env(backend=None, single_rank_group=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)
    and returns (rank, world_size, local_rank).  World size 1 needs no group;
    `single_rank_group` creates one anyway (a launcher started this process:
    `torch.distributed.run --nproc-per-node 1` then runs every collective of
    the N-rank path over RCCL on the one device)."""
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        if backend is None:
            # RCCL needs one distinct GPU per rank; with fewer devices than
            # ranks (dry runs on a 1-GPU box) fall back to gloo
            backend = "nccl" if (torch.cuda.is_available() and
                                 torch.cuda.device_count() >= world) else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if torch.cuda.is_available() and torch.cuda.device_count() < world:
            # ranks share devices (dry runs on a small box): one HIP queue per
            # process -- the chain's second stream makes co-tenants on one GPU
            # time-slice badly (measured 3x slower), see chain.hip chain_step
            os.environ.setdefault("BBX_CHAIN_FORK", "0")
            # ... and their device-heavy set-up (design generation, the
            # library's transposition sorts) goes one rank at a time: eight
            # concurrent 1e8-entry sorts from eight processes do not finish
            # (setup_turn below; libbbx honours BBX_SETUP_LOCK itself)
            os.environ.setdefault(
                "BBX_SETUP_LOCK", "/tmp/bbx_setup_%s.lock"
                % os.environ.get("MASTER_PORT", "0"))
        if backend == "nccl":
            # RCCL binds a communicator to the current device
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class setup_turn():
    """`with chains.setup_turn(): ...` around device-heavy set-up code of a
    rank: a no-op unless BBX_SETUP_LOCK names a lock file (ranks sharing a
    GPU, see init_process_group_from_env), then the library's process-global,
    RE-ENTRANT lock on it (`bbx_setup_lock_acquire`): the design constructors
    take the same lock around their own device work, so a constructor may be
    called inside the bracket.  The device is synchronised before the turn
    is handed on."""

    def __enter__(self):
        self._held = 0
        if os.environ.get("BBX_SETUP_LOCK"):
            from . import _lib
            self._held = _lib.load().bbx_setup_lock_acquire()
        return self

    def __exit__(self, *exc):
        if self._held > 0:
            try:
                import torch
                if torch.cuda.is_available():
                    torch.cuda.synchronize()
            except Exception:      # noqa: BLE001
                pass
            from . import _lib
            _lib.load().bbx_setup_lock_release()
            self._held = 0
        return False


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
    if not (dist.is_available() and dist.is_initialized()):
        return local.unsqueeze(0)
    world = dist.get_world_size()
    rank = dist.get_rank()
    local = local.contiguous()
    came_from = local.device
    if dist.get_backend() == "gloo" and local.is_cuda:
        local = local.cpu()
    elif dist.get_backend() == "nccl" and not local.is_cuda:
        local = local.cuda()          # RCCL moves device memory only
    if rank == dst:
        bucket = [torch.empty_like(local) for _ in range(world)]
        dist.gather(local, gather_list=bucket, dst=dst)
        return torch.stack(bucket).to(came_from)
    dist.gather(local, gather_list=None, dst=dst)
    return None


def max_over_ranks(value):
    """MAX all-reduce of a Python float (timing: the slowest rank counts)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
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


def run_chains(bridge, n_chain, n_iter, n_burnin=0, thin=1, seed=0,
               init=None, params_to_save=('coef', 'global_scale', 'logp'),
               options=None, dst=0, batch=False):
    """`n_chain` independent chains of `bridge.gibbs(...)` (BASELINE config 5).

    Inside a torch.distributed.run launch (one rank per GPU, each with its own
    replica of the design) rank r runs chains r, r + world, ...; chain k uses
    seed `seed + k` whatever the number of ranks.  Nothing is exchanged while
    sampling; at the end each saved parameter is gathered ONCE on rank `dst`.

    batch: what a rank does with SEVERAL chains.
      False (default)  one after the other through `bridge.gibbs`: chain k's
                       samples are bit-for-bit the same whatever the number of
                       ranks (8 chains on 8, 2 or 1 GPU) -- the reproducibility
                       contract of this function;
      'auto'           in batches that share every pass over X
                       (`bridge.gibbs_batch`), of the width
                       `bridge.batch_width` picks from the library's cost model
                       and the free GPU memory -- faster (1.35x for sparse pairs
                       at 1M x 50k, 6.7x for 16 dense chains), but a batched chain
                       equals its single run only to rounding, and an MCMC
                       trajectory diverges from rounding differences: the
                       samples then DEPEND on how the chains were grouped, i.e.
                       on the number of ranks and on the chosen width;
      int k            batches of exactly k where k chains are left (k = 2, 4
                       sparse; 2 ... 32 dense), the rest alone; same caveat.
                       Built even where the library's cost model prices the
                       width below single chains (`allow_slow`).
    The decision is recorded per chain in mcmc_info['batch'] = {'requested':
    batch, 'width': w, 'slot': i} (width 1: run alone).

    Returns on `dst` (samples, infos): samples[name] has the chain index first
    and the MCMC index last, e.g. 'coef' is (n_chain, P, n_sample); infos is
    the list of this rank's mcmc_info dicts (chain index in info['chain']).
    Other ranks get (None, infos).  Without a process group all chains run one
    after the other on this process's GPU.
    """
    import copy
    import torch
    rank, world, _ = init_process_group_from_env()
    mine = split_chains(n_chain, world, rank)
    per_rank = -(-n_chain // world)               # ranks pad to equal counts
    init = {'global_scale': .1} if init is None else init
    if params_to_save == 'all':
        params_to_save = ('coef', 'local_scale', 'global_scale', 'logp',
                          'obs_prec')
    n_sample = (n_iter - n_burnin) // thin
    kept, infos = {}, []
    # Vector-valued samples never visit the host on their way to the
    # collective: a chain writes them (bbx_chain_run's device buffers) into
    # its slot of ONE device tensor per parameter, [per_rank, n_sample, dim],
    # and that tensor is what RCCL gathers.  (A bridge without the hook -- the
    # CPU stand-in of the gloo tests -- and batched chains hand over host
    # arrays, which are put into the same slots.)
    dev_names = ()
    rng_mode = options.get('rng', 'device') if isinstance(options, dict) \
        else getattr(options, 'rng', 'device')
    if getattr(bridge, 'supports_device_samples', False) \
            and torch.cuda.is_available() and rng_mode != 'reference':
        dev_names = tuple(k for k in ('coef', 'local_scale', 'obs_prec')
                          if k in params_to_save)
    slabs = {}
    if dev_names:
        design = bridge.model.design
        dims = {'coef': bridge.n_pred,
                'local_scale': bridge.n_pred - bridge.n_unshrunk,
                'obs_prec': bridge.n_obs if bridge.model.name == 'logit'
                else 1}
        for name in dev_names:
            slabs[name] = torch.zeros(
                (per_rank, n_sample, dims[name]), dtype=torch.float64,
                device='cuda:%d' % design.device)

    def keep(k, samples, info):
        slot = len(infos)
        info['chain'] = k
        infos.append(info)
        samples = dict(samples)
        samples['n_cg_iter'] = info['_reg_coef_sampling_info']['n_cg_iter']
        for name, arr in samples.items():
            arr = np.asarray(arr, dtype=np.float64)
            if name in slabs:     # host copy of a batched chain: into its slot
                slabs[name][slot].copy_(torch.from_numpy(
                    np.ascontiguousarray(arr.T).reshape(n_sample, -1)))
            else:
                kept.setdefault(name, []).append(arr)
    # batch != False: this rank's chains go through the design in BATCHES that
    # share every pass over X (BayesBridge.gibbs_batch) instead of one after
    # the other; what does not fill a batch runs alone.  A chain's seed (hence
    # its Philox streams) is seed + k either way; batched and single runs of a
    # chain agree to rounding, not bit for bit.
    if batch not in (False, None, 'auto') and not (
            isinstance(batch, int) and batch >= 2):
        raise ValueError("batch must be False, 'auto' or a width >= 2")
    todo = list(mine)
    width_of = getattr(bridge, 'batch_width', None) if batch == 'auto' else None

    def pick_width(n_left):
        if batch == 'auto':
            return width_of(n_left, params_to_save, options) if width_of else 0
        if batch and n_left >= int(batch):
            return int(batch)
        return 0
    while todo:
        width = pick_width(len(todo))
        if width >= 2:
            group, todo = todo[:width], todo[width:]
            # (an explicit integer width is the caller's decision: it also
            # builds a width the library's cost model prices below single
            # chains, e.g. four sparse chains at 1M x 50k; 'auto' never asks
            # for such a width)
            results = bridge.gibbs_batch(
                [chain_seed(seed, k) for k in group], n_iter, n_burnin, thin,
                init=copy.deepcopy(init), params_to_save=params_to_save,
                options=options, allow_slow=batch != 'auto')
            for k, (samples, info) in zip(group, results):
                info['batch']['requested'] = batch
                keep(k, samples, info)
        else:
            k = todo.pop(0)
            extra = {}
            if slabs:
                extra['_device_out'] = {name: slab[len(infos)]
                                        for name, slab in slabs.items()}
            samples, info = bridge.gibbs(
                n_iter, n_burnin, thin, seed=chain_seed(seed, k),
                init=copy.deepcopy(init), params_to_save=params_to_save,
                coef_sampler_type='cg', options=options, **extra)
            info['batch'] = {'requested': batch or False, 'width': 1, 'slot': 0}
            keep(k, samples, info)
    names = sorted(kept) if kept else None
    if world > 1:                                 # ranks without a chain
        import torch.distributed as dist
        box = [names]
        src_rank = 0                              # rank 0 always owns chain 0
        dist.broadcast_object_list(box, src=src_rank)
        names = box[0]
    merged = {}

    def place(got, shape):
        # [world, per_rank, ...] -> [n_chain, ...] in chain order
        out = np.empty((n_chain,) + tuple(shape))
        for r in range(got.shape[0]):
            for slot, k in enumerate(split_chains(n_chain, got.shape[0], r)):
                out[k] = got[r, slot]
        return out
    # (1) the device slabs: straight into the collective
    for name in dev_names:
        got = gather_chain_samples(slabs[name], dst=dst)
        if got is None:
            continue
        got = got.cpu().numpy()                   # [world, per_rank, n_sample, dim]
        out = place(got, got.shape[2:])
        out = np.ascontiguousarray(np.swapaxes(out, 1, 2))  # MCMC index last
        if name == 'obs_prec' and out.shape[1] == 1 \
                and bridge.model.name == 'linear':
            out = out[:, 0]
        merged[name] = out
    # (2) whatever lives on the host (the per-sample scalars; everything for a
    # bridge without device buffers), one gather per parameter
    for name in names:
        have = kept.get(name, [])
        if have:
            block = np.stack(have)
        else:
            block = None
        shape = None if block is None else block.shape[1:]
        if world > 1:
            import torch.distributed as dist
            box = [shape]
            dist.broadcast_object_list(box, src=0)
            shape = tuple(box[0])
        padded = np.zeros((per_rank,) + tuple(shape))
        if block is not None:
            padded[:len(have)] = block
        got = gather_chain_samples(torch.from_numpy(padded), dst=dst)
        if got is None:
            continue
        merged[name] = place(got.numpy(), shape)
    if world > 1 and rank != dst:
        return None, infos
    return merged, infos


def merge_chain_outputs(gathered, params=('coef',)):
    """[world, n_sample, P] -> dict of NumPy arrays with the chain index first
    and, as in the reference's `samples`, the MCMC index last."""
    arr = gathered.cpu().numpy()
    return {'coef': np.transpose(arr, (0, 2, 1))}
