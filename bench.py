#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the CG-accelerated sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full Gibbs iteration (beta by prior-preconditioned CG,
Omega, tau, lambda, log posterior) of a logit model on the synthetic binary
design of BASELINE.json's headline config (1M x 50k, nnz ~ 1e8), everything
resident in HBM.  For N > 1 every rank runs its own chain on a full replica of
X (weak scaling; seeds 111 + rank) and the kept coefficient samples are
gathered on rank 0 over RCCL once, inside the timed region.

Prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (ROOT, os.path.join(ROOT, "bayes-bridge_amd")):
    if path not in sys.path:
        sys.path.insert(0, path)

CONFIGS = {
    # name: (n, p, binary_pred_freq)      sparse binary logit designs
    "config2": (100000, 10000, .01),
    "config3": (1000000, 50000, .002),
    # BASELINE config 4: linear model, dense N(0,1) design stored in f32
    "config4": (200000, 8000, None),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="config3", choices=sorted(CONFIGS))
    ap.add_argument("--storage", default="auto",
                    choices=["auto", "csr", "tiled"])
    ap.add_argument("--cpu-baseline-iters", type=int, default=2,
                    help="Gibbs iterations of the CPU oracle (0 = skip)")
    ap.add_argument("--seed", type=int, default=111)
    return ap.parse_args()


def build_dense_problem(torch, cfg, seed, device):
    """Config 4: X ~ N(0,1) in f32 from torch's Philox generator (seed 111),
    y = X beta + N(0,1) with the demo's beta (demo.ipynb cell 5)."""
    n, p, _ = CONFIGS[cfg]
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    X = torch.randn((n, p), generator=gen, device=device, dtype=torch.float32)
    offset = X.double().mean(dim=0)
    beta = torch.zeros(15, dtype=torch.float64, device=device)
    beta[:5], beta[5:10], beta[10:15] = 1.5, 1., .5
    y = X[:, :15].double() @ beta + torch.randn(
        n, generator=gen, device=device, dtype=torch.float64)
    return dict(n=n, p=p, nnz=n * (p + 1), X=X, offset=offset, y=y)


def build_problem(torch, cfg, seed, device):
    """Synthetic design + logit outcome, generated in HBM."""
    from bayesbridge_amd import simulate
    n, p, freq = CONFIGS[cfg]
    indptr, indices = simulate.simulate_binary_csr_device(
        n, p, freq, seed=seed, device=device)
    nnz = int(indices.numel())
    col_count = torch.bincount(indices.long(), minlength=p)
    offset = col_count.double() / n                 # column means of a 0/1 X
    beta = torch.zeros(p, dtype=torch.float64, device=device)
    beta[:5], beta[5:10], beta[10:15] = 1.5, 1., .5  # demo.ipynb cell 5
    # eta = X beta (only 15 non-zero coefficients)
    rows = torch.repeat_interleave(
        torch.arange(n, device=device), (indptr[1:] - indptr[:-1]).long())
    eta = torch.zeros(n, dtype=torch.float64, device=device)
    eta.index_add_(0, rows, beta[indices.long()])
    del rows
    gen = torch.Generator(device=device)
    gen.manual_seed(1)                               # simulate_outcome seed=1
    prob = torch.sigmoid(eta)
    n_success = (torch.rand(n, generator=gen, device=device,
                            dtype=torch.float64) < prob).double()
    return dict(n=n, p=p, nnz=nnz, indptr=indptr, indices=indices,
                offset=offset, n_success=n_success)


def cpu_baseline(torch, prob, state, n_iters, seed):
    """Times the CPU oracle (SciPy CSR products + SciPy cg, the primitives the
    reference runs; one thread) on the SAME matrix, started from the GPU
    chain's current state, for `n_iters` Gibbs iterations."""
    import numpy as np
    import scipy.sparse as sparse
    from oracle.gibbs import OracleGibbs
    from oracle.summarizer import CoefSummarizer
    n, p = prob["n"], prob["p"]
    X = sparse.csr_matrix(
        (np.ones(prob["nnz"]), prob["indices"].cpu().numpy(),
         prob["indptr"].cpu().numpy()), shape=(n, p))
    n_success = prob["n_success"].cpu().numpy()
    chain = OracleGibbs((n_success, np.ones(n)), X, 'logit',
                        bridge_exponent=.5, regularizing_slab_size=2.,
                        use_scipy_cg=True)
    from oracle.rng import OracleRandom
    chain.rng = OracleRandom(seed)
    coef, obs_prec, lscale, gscale, mean, square, n_avg = state
    summ = CoefSummarizer(chain.P, chain.nu, chain.slab)
    summ.set_state({'mean': mean, 'square': square, 'n_averaged': n_avg})
    n_cg = []
    t0 = time.perf_counter()
    for _ in range(n_iters):
        coef, info = chain.draw_coef(obs_prec, gscale, lscale, summ)
        obs_prec = chain.draw_obs_prec(coef)
        gscale = chain.draw_gscale(coef[chain.nu:])
        lscale = chain.draw_lscale(gscale, coef[chain.nu:])
        chain.logp(coef, gscale, obs_prec)
        n_cg.append(info['n_iter'])
    dt = time.perf_counter() - t0
    return dict(value=n_iters / dt, unit="Gibbs iters/sec", cores=1,
                kind="port",
                sample="%d Gibbs iterations of the NumPy/SciPy oracle "
                       "(scipy.sparse CSR @, .T @, scipy.sparse.linalg.cg; C "
                       "Polya-Gamma/tilted-stable) on the same %dx%d nnz=%d "
                       "design from the GPU chain's post-warm-up state; mean "
                       "n_cg=%.1f; %.1f s" % (
                           n_iters, n, p, prob["nnz"],
                           float(np.mean(n_cg)), dt),
                host_cores=os.cpu_count())


def committed_traffic(design, which, cfg):
    """HBM bytes per launch of the dominant kernel from the committed PMC
    passes (profiles/r01_spmv_profile.json; FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE), matched by launch grid; None when no profile of
    this workload/geometry is committed."""
    if cfg != "config3" or design.storage_format != "tiled":
        return None
    path = os.path.join(ROOT, "profiles", "r01_spmv_profile.json")
    if not os.path.exists(path):
        return None
    info = design.tiled_info()["X" if which == "dot" else "Xt"]
    n, P = design.shape
    rows = n if which == "dot" else P - 1
    n_wg = -(-rows // info["PR"]) * info["G"]
    with open(path) as fh:
        prof = json.load(fh)
    entry = prof.get("hbm_traffic", {}).get("grid=%d" % n_wg)
    return int(entry["total_bytes"]) if entry else None


def main():
    args = parse_args()
    # stdout carries exactly ONE line (the JSON result of rank 0): libraries
    # that write to file descriptor 1 (gloo's connection notes, rocm tools)
    # are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    from ctypes import byref, c_double, c_int64, c_void_p
    from bayesbridge_amd import (BayesBridge, HipSparseDesignMatrix, _lib,
                                 chains)
    rank, world, local_rank = chains.init_process_group_from_env()
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus must equal WORLD_SIZE")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % max(n_dev, 1)   # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    lib = _lib.load()

    dense = args.config == "config4"
    chain = c_void_p()
    import math
    unit = math.gamma(2 / .5) / math.gamma(1 / .5)   # prior.py:163-167
    sd_unshrunk = np.array([np.inf])
    if dense:
        from bayesbridge_amd.design_matrix import (HipDenseDesignMatrix,
                                                   HipDesignMatrix)
        prob = build_dense_problem(torch, args.config, args.seed, device)
        n, p, nnz = prob["n"], prob["p"], prob["nnz"]
        design = HipDenseDesignMatrix.__new__(HipDenseDesignMatrix)
        HipDesignMatrix.__init__(design)
        design.centered, design.intercept_added = True, True
        design.column_offset = None
        torch.cuda.synchronize()
        _lib.check(lib.bbx_design_create_dense_dev(
            n, p, c_void_p(prob["X"].data_ptr()), _lib.F32, _lib.F32,
            c_void_p(prob["offset"].data_ptr()), 1, dev_index,
            byref(design._h)))
        del prob["X"]
        P = p + 1
        outcome = prob["y"].cpu().numpy()
        _lib.check(lib.bbx_chain_create(
            design.handle, _lib.MODEL_LINEAR,
            outcome.ctypes.data_as(c_void_p), None, 1,
            sd_unshrunk.ctypes.data_as(c_void_p), .5, 2., 0., 0.,
            chains.chain_seed(args.seed, rank), byref(chain)))
        coef0 = np.zeros(P)
        coef0[0] = outcome.mean()
    else:
        prob = build_problem(torch, args.config, args.seed, device)
        n, p, nnz = prob["n"], prob["p"], prob["nnz"]
        torch.cuda.synchronize()
        design = HipSparseDesignMatrix.from_device_csr(
            n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(),
            None, prob["offset"].data_ptr(), add_intercept=True,
            device=dev_index, storage=args.storage)
        P = p + 1
        # chain: prior and init of the reference demo (demo.ipynb cells 7, 9)
        n_success = prob["n_success"].cpu().numpy()
        _lib.check(lib.bbx_chain_create(
            design.handle, _lib.MODEL_LOGIT,
            n_success.ctypes.data_as(c_void_p), None, 1,
            sd_unshrunk.ctypes.data_as(c_void_p), .5, 2., 0., 0.,
            chains.chain_seed(args.seed, rank), byref(chain)))
        coef0 = np.zeros(P)
        ph = n_success.mean()
        coef0[0] = math.log(ph / (1 - ph))               # intercept MLE
    lscale0 = np.ones(P - 1) * unit
    g0 = c_double(.01 / unit)                        # init global_scale=.01
    _lib.check(lib.bbx_chain_set_state(
        chain, coef0.ctypes.data_as(c_void_p), None,
        lscale0.ctypes.data_as(c_void_p), byref(g0)))
    _lib.check(lib.bbx_chain_init_obs_prec(chain))

    K, W = args.steps, args.warmup
    ncg_w = np.zeros(max(W, 1))
    if W > 0:
        _lib.check(lib.bbx_chain_run(chain, W, 0, 1, 500, 0., None, None, None,
                                     None, None,
                                     ncg_w.ctypes.data_as(c_void_p)))
    # state after warm-up (for the CPU baseline)
    state = None
    if rank == 0 and world == 1 and args.cpu_baseline_iters > 0 and not dense:
        coef = np.empty(P)
        obs = np.empty(n)
        ls = np.empty(P - 1)
        g = c_double()
        _lib.check(lib.bbx_chain_get_state(
            chain, coef.ctypes.data_as(c_void_p),
            obs.ctypes.data_as(c_void_p), ls.ctypes.data_as(c_void_p),
            byref(g)))
        mean, square, navg = np.empty(P), np.empty(P), c_int64()
        _lib.check(lib.bbx_chain_get_summary(
            chain, mean.ctypes.data_as(c_void_p),
            square.ctypes.data_as(c_void_p), byref(navg)))
        state = (coef, obs, ls, float(g.value), mean, square,
                 int(navg.value))

    d_coef = torch.empty((max(K, 1), P), dtype=torch.float64, device=device)
    gs, lp, ncg = np.zeros(max(K, 1)), np.zeros(max(K, 1)), np.zeros(max(K, 1))
    # HIP events around one dot/Tdot launch in 16 (timing every launch costs
    # ~10% of the iteration; DESIGN.md "Measurement")
    design.set_timing(True, every=16)
    design.reset_timing()
    if world > 1:
        # part of the warm-up: the first gather of a process group sets up the
        # point-to-point connections (RCCL does that lazily, 100s of ms)
        chains.gather_chain_samples(d_coef, dst=0)
    chains.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(lib.bbx_chain_run(
        chain, K, 0, 1, 500, 0., c_void_p(d_coef.data_ptr()), None, None,
        gs.ctypes.data_as(c_void_p), lp.ctypes.data_as(c_void_p),
        ncg.ctypes.data_as(c_void_p)))
    gathered = chains.gather_chain_samples(d_coef, dst=0)
    chains.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = chains.max_over_ranks(elapsed)
    timing = design.get_timing()
    design.set_timing(False)

    if rank == 0:
        assert gathered is not None and gathered.shape[0] == world
        assert bool(torch.isfinite(gathered).all())
        dot_b, tdot_b = design.matvec_bytes
        per = {}
        for name, b in (("dot", dot_b), ("tdot", tdot_b)):
            cnt, ms = timing[name]
            avg_ms = ms / max(cnt, 1)
            per[name] = dict(launches=cnt, avg_ms=avg_ms, bytes=b,
                             gbs=b / avg_ms / 1e6 if avg_ms > 0 else 0.)
        dom = "dot" if timing["dot"][1] >= timing["tdot"][1] else "tdot"
        ach = per[dom]["gbs"]
        traffic = committed_traffic(design, dom, args.config)
        # measured on this box, same size as one launch's algorithmic bytes
        # and at 2 GB: what a plain streaming kernel reaches
        probe_small = _lib.hbm_probe(int(per[dom]["bytes"]), 50, dev_index)
        probe_large = _lib.hbm_probe(2 << 30, 10, dev_index)
        roofline = dict(
            bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS,
            unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
            stream_probe_gbs={
                "read_same_bytes": round(probe_small[0], 1),
                "copy_same_bytes": round(probe_small[1], 1),
                "read_2GiB": round(probe_large[0], 1),
                "copy_2GiB": round(probe_large[1], 1)},
            kernel=dom + " (" + design.storage_format + ")",
            avg_launch_ms=round(per[dom]["avg_ms"], 5),
            algorithmic_bytes_per_launch=per[dom]["bytes"],
            other={k: dict(avg_ms=round(v["avg_ms"], 5),
                           gbs=round(v["gbs"], 1), launches=v["launches"])
                   for k, v in per.items()})
        line = {
            "metric": "Gibbs iters/sec (cg sampler)",
            "value": round(world * K / elapsed, 4),
            "unit": "Gibbs iters/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(1e3 * elapsed / K, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    "%s: linear, dense N(0,1) %dx%d stored f32 (torch Philox "
                    "seed %d), centred + intercept, one independent chain "
                    "per GPU, seeds %d+rank" % (args.config, n, p, args.seed,
                                                args.seed)) if dense else (
                    "%s: logit, sparse binary CSR %dx%d nnz=%d "
                    "(simulate_data.py distribution, f=%g), one "
                    "independent chain per GPU, seeds %d+rank"
                    % (args.config, n, p, nnz, CONFIGS[args.config][2],
                       args.seed)),
                "storage": design.storage_format,
                "mean_n_cg_iter": round(float(ncg[:K].mean()), 2),
                "mean_n_cg_iter_warmup": round(float(ncg_w[:W].mean()), 2)
                if W > 0 else None,
                "parallelism": "chains=%d" % world,
                "devices": min(world, n_dev),
            },
            "roofline": roofline,
        }
        if state is not None:
            line["cpu_baseline"] = cpu_baseline(
                torch, prob, state, args.cpu_baseline_iters, args.seed)
        else:
            line["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    lib.bbx_chain_destroy(chain)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
