#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the CG-accelerated sampler on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full Gibbs iteration (beta by prior-preconditioned CG,
Omega, tau, lambda, log posterior) of a logit model on the synthetic binary
design of BASELINE.json's headline config (1M x 50k, nnz ~ 1e8), everything
resident in HBM.  For N > 1 every rank runs its own chain on a full replica of
X (weak scaling; seeds 111 + rank) and the kept coefficient samples are
gathered on rank 0 over RCCL once, inside the timed region.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment launches the N
ranks itself (`python -m torch.distributed.run ... bench.py ...` as a child
process, before this process touches torch or HIP) and relays rank 0's line.

Prints ONE JSON line (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# (BBX_PACKAGE_DIR: another build of the package, for A/B runs of two libraries)
for path in (ROOT, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd"))):
    if path not in sys.path:
        sys.path.insert(0, path)

CONFIGS = {
    # name: (n, p, binary_pred_freq)      sparse binary logit designs
    "config2": (100000, 10000, .01),
    "config3": (1000000, 50000, .002),
    # BASELINE config 4: linear model, dense N(0,1) design stored in f32
    "config4": (200000, 8000, None),
    # small smoke configurations (launcher dry runs, tests)
    "tiny": (20000, 1000, .02),
    "tiny-dense": (6000, 400, None),
    # between configs 2 and 3 (A/Bs of size-dependent choices)
    "mid": (400000, 20000, .005),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
ALPHA, SLAB = .5, 2.   # demo.ipynb cell 7 prior


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--burnin", type=int, default=None,
                    help="untimed Gibbs iterations run as part of the chain's "
                         "initialisation (the reference runs an L-BFGS mode "
                         "search there, bayesbridge.py:333) so that the chain "
                         "is past its transient when the W warm-up steps "
                         "begin: tau, logp and n_cg are stationary after "
                         "~250-300 iterations at config 3 "
                         "(profiles/r02_ncg_trajectory.txt).  Default: 300 for "
                         "the sparse configs, 100 for config4 (n_cg still falls "
                         "by 6 %% over iterations 10-60 there)")
    ap.add_argument("--config", default="config3", choices=sorted(CONFIGS))
    ap.add_argument("--storage", default="auto",
                    choices=["auto", "csr", "tiled"])
    ap.add_argument("--cpu-baseline-iters", type=int, default=None,
                    help="Gibbs iterations of each CPU baseline (0 = skip; "
                         "default 5, 2 for the dense configs: one config-4 "
                         "iteration is ~110 operator applications of two "
                         "12.8 GB dgemv each)")
    ap.add_argument("--seed", type=int, default=111)
    ap.add_argument("--live-traffic", type=int, default=1, choices=[0, 1],
                    help="rank 0 at N = 1, sparse configs 2 / 3: measure "
                         "`roofline.traffic` in this run by two rocprofv3 --pmc "
                         "child passes over the product kernels (adds ~40 s "
                         "after the timed region); 0: the committed profile's "
                         "figure")
    ap.add_argument("--dense-storage", default="float32",
                    choices=["float32", "float64"],
                    help="config4 only: storage type of the dense matrix "
                         "(BASELINE: float32; float64 is the reference's own "
                         "type, 12.8 GB)")
    ap.add_argument("--multi-chain", default=None,
                    help="comma-separated batch widths for the `multi_chain` "
                         "object (k chains on ONE GPU sharing every pass over "
                         "X; rank 0 at N = 1 only; '0' = skip).  Default: 2,4 "
                         "for the sparse configs, 4,8,16,32 for config4")
    ap.add_argument("--multi-chain-steps", type=int, default=20)
    ap.add_argument("--cg-fold", type=int, default=None, choices=[0, 1],
                    help="A/B: force the 3-launch CG iteration (direction step "
                         "inside the X~ v kernel, bbx_design_set_cg_fold) on / "
                         "off; default: the library's rule (on up to 250 000 "
                         "rows)")
    ap.add_argument("--timing-blocks", default="all",
                    choices=["first", "all", "none"],
                    help="which of the `repeat` blocks run with the kernel "
                         "stamps on (LABNOTES R5.1: a stamped launch costs ~8 us; "
                         "with 'first' block 0 alone paid for them, 4 %% at "
                         "config 2); the line's `roofline` is measured in block "
                         "0 and needs 'first' or 'all'")
    ap.add_argument("--timing-every", type=int, default=0,
                    help="one launch in N carries kernel stamps; 0 (default): "
                         "chosen after the warm-up so that a K-step block "
                         "holds ~24 stamped launches per kernel, between one "
                         "in 8 and one in 64 (a stamped launch costs ~8 us: "
                         "LABNOTES R5.11)")
    ap.add_argument("--repeat", type=int, default=5,
                    help="how many times the K-step block is run in all for "
                         "the `repeat` object (the first is the timed region "
                         "behind `value`; 1 = no extra blocks)")
    return ap.parse_args(argv)


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


def cpu_baselines(prob, state, n_iters, seed):
    """Times the CPU oracle on the SAME matrix, started from the GPU chain's
    post-warm-up state, for `n_iters` Gibbs iterations each:
      port      SciPy CSR products + scipy.sparse.linalg.cg -- the primitives
                the reference runs (single-threaded, like SciPy's SpMV)
      port-omp  the same chain with OpenMP products and CG on every core of
                the affinity mask (oracle/csrc/oracle_cg_omp.cpp: value-free
                int32 CSR, each thread's rows first-touched by that thread) and
                the scalar samplers on a thread pool with per-block PCG64
                streams; reports the products' effective GB/s beside it."""
    import numpy as np
    import scipy.sparse as sparse
    from oracle.gibbs import OracleGibbs
    from oracle.rng import OracleRandom
    from oracle.summarizer import CoefSummarizer
    n, p = prob["n"], prob["p"]
    X = sparse.csr_matrix(
        (np.ones(prob["nnz"]), prob["indices"].cpu().numpy(),
         prob["indptr"].cpu().numpy()), shape=(n, p))
    n_success = prob["n_success"].cpu().numpy()
    coef0, obs0, ls0, g0, mean, square, n_avg = state
    from oracle.omp_baseline import cpu_quota, usable_cores
    from oracle.rng import ParallelOracleRandom
    host_cores = os.cpu_count()
    # every core this container can keep busy: the affinity mask capped by the
    # cgroup's CPU quota (threads beyond it are throttled, not run)
    omp_cores = usable_cores()

    def run(kind, cores, **kw):
        chain = OracleGibbs((n_success, np.ones(n)), X, 'logit',
                            bridge_exponent=ALPHA,
                            regularizing_slab_size=SLAB, **kw)
        extra = {}
        if kind == "port":
            chain.rng = OracleRandom(seed)
        else:
            # scalar samplers on a thread pool with per-block PCG64 streams
            chain.rng = ParallelOracleRandom(seed, min(cores, 64))
            # effective rate of the products alone (value-free int32 CSR,
            # SURVEY 8(d)'s bytes of the format actually read)
            d = chain.design
            v, w = np.ones(d.shape[1]), np.ones(d.shape[0])
            d.dot(v), d.Tdot(w)
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                d.dot(v)
            t1 = time.perf_counter()
            for _ in range(reps):
                d.Tdot(w)
            t2 = time.perf_counter()
            db, tb = d.product_bytes
            extra = dict(
                dot_gbs=round(db * reps / (t1 - t0) / 1e9, 1),
                tdot_gbs=round(tb * reps / (t2 - t1) / 1e9, 1),
                cpu_quota=cpu_quota(),
                product_format="value-free int32 CSR of X and of X^T, each "
                               "thread's rows first-touched by that thread")
        summ = CoefSummarizer(chain.P, chain.nu, chain.slab)
        summ.set_state({'mean': mean, 'square': square, 'n_averaged': n_avg})
        coef, obs_prec, lscale, gscale = coef0, obs0, ls0, g0
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
        what = ("scipy.sparse CSR @, .T @, scipy.sparse.linalg.cg; C "
                "Polya-Gamma/tilted-stable samplers on 1 thread"
                if kind == "port" else
                "OpenMP value-free CSR / CSR-of-X^T products (NUMA first "
                "touch) and CG loop in C++ on %d threads; C samplers on a "
                "%d-thread pool" % (cores, min(cores, 64)))
        out = dict(value=round(n_iters / dt, 5), unit="Gibbs iters/sec",
                   cores=cores, kind=kind,
                   sample="%d Gibbs iterations of the CPU oracle chain (%s) "
                          "on the same %dx%d nnz=%d design from the GPU "
                          "chain's post-warm-up state; mean n_cg=%.1f; "
                          "%.1f s" % (n_iters, what, n, p, prob["nnz"],
                                      float(np.mean(n_cg)), dt),
                   host_cores=host_cores)
        out.update(extra)
        return out
    port = run("port", 1, use_scipy_cg=True)
    omp = run("port-omp", omp_cores, omp_threads=omp_cores)
    return port, omp


def cpu_baseline_dense(x_host, offset, y, state, n_iters, seed):
    """Config 4's CPU leg: the oracle chain on what the reference's dense path
    runs -- `X.dot(v)` / `X.T.dot(w)` of a C-ordered float64 array (two BLAS
    dgemv per operator application, dense_matrix.py:42,52) and
    scipy.sparse.linalg.cg -- for `n_iters` Gibbs iterations from the GPU
    chain's post-warm-up state.  X is the SAME matrix (the f32-representable
    entries the GPU generated), centred, with the intercept column, in f64 as
    the reference holds it; BLAS threads = the cores this container can keep
    busy (threadpoolctl), stated in `cores`."""
    import numpy as np
    from threadpoolctl import threadpool_info, threadpool_limits
    from oracle.design_matrix import OracleDenseDesign
    from oracle.gibbs import OracleGibbs
    from oracle.omp_baseline import usable_cores
    from oracle.rng import OracleRandom
    from oracle.summarizer import CoefSummarizer
    n, p = x_host.shape
    cores = usable_cores()
    t0 = time.perf_counter()
    Xf = np.empty((n, p + 1))
    Xf[:, 0] = 1.
    step = max(1, (64 << 20) // (8 * p))
    for r0 in range(0, n, step):       # centre while widening, chunk by chunk
        Xf[r0:r0 + step, 1:] = x_host[r0:r0 + step]
        Xf[r0:r0 + step, 1:] -= offset
    prep_s = time.perf_counter() - t0
    coef0, obs0, ls0, g0, mean, square, n_avg = state
    with threadpool_limits(limits=cores):
        blas = sorted({"%s %s" % (i.get("internal_api"), i.get("num_threads"))
                       for i in threadpool_info()
                       if i.get("user_api") == "blas"})
        chain = OracleGibbs(y, OracleDenseDesign.from_full(Xf), 'linear',
                            bridge_exponent=ALPHA, regularizing_slab_size=SLAB,
                            use_scipy_cg=True)
        chain.rng = OracleRandom(seed)
        summ = CoefSummarizer(chain.P, chain.nu, chain.slab)
        summ.set_state({'mean': mean, 'square': square, 'n_averaged': n_avg})
        d = chain.design
        v, w = np.ones(p + 1), np.ones(n)
        d.dot(v), d.Tdot(w)
        t1 = time.perf_counter()
        d.dot(v)
        t2 = time.perf_counter()
        d.Tdot(w)
        t3 = time.perf_counter()
        coef, obs_prec, lscale, gscale = coef0, obs0, ls0, g0
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
    return dict(
        value=round(n_iters / dt, 5), unit="Gibbs iters/sec", cores=cores,
        kind="port",
        sample="%d Gibbs iterations of the CPU oracle chain (NumPy X.dot / "
               "X.T.dot = BLAS dgemv on a C-ordered float64 %dx%d array, "
               "scipy.sparse.linalg.cg, linear model) on the same matrix "
               "from the GPU chain's post-warm-up state; mean n_cg=%.1f; "
               "%.1f s (+ %.1f s to widen and centre the matrix)"
               % (n_iters, n, p + 1, float(np.mean(n_cg)), dt, prep_s),
        blas=blas, host_cores=os.cpu_count(),
        dot_gbs=round(8. * n * (p + 1) / (t2 - t1) / 1e9, 1),
        tdot_gbs=round(8. * n * (p + 1) / (t3 - t2) / 1e9, 1))


def iteration_bytes(mean_ncg, op_bytes, dot_bytes, tdot_bytes, n, P,
                    dense_single_pass=False, vec_passes=15):
    """Algorithmic bytes of ONE whole Gibbs iteration (`roofline.iteration`):
    n_cg operator applications; for the warm start one more application (dense
    single-pass kernel) or one product with X~ plus ONE with X~^T for the
    initial residual (cg_sampler.hip TD_RESID); the linear predictor of the
    Omega update; the Omega vector every operator application scales by (8 n
    bytes: the product byte counts hold the vector in and out only); 15
    P-vector passes per CG iteration (direction kernel: r, p, s, offset, d in,
    p, s.*p out; Tdot epilogue: offset, p, d, s, x, r in, x, r out -- its slab
    read is part of tdot_bytes; vec_passes = 17 where the direction step rides
    in the X~ v kernel: a second slice vector, s.*r written and read), ~64
    bytes per row and ~30 P-vector passes for the eta draws and the chain
    kernels.  Only bytes that are moved are
    credited.  Pinned against the PMC counters of a profiled chain: 0.986 of
    the measured HBM traffic (profiles/r03_iteration_traffic.json,
    tests/test_bench_byte_model.py)."""
    if dense_single_pass:
        total = (mean_ncg + 1) * op_bytes + dot_bytes
    else:
        total = mean_ncg * op_bytes + 2 * dot_bytes + tdot_bytes
    total += (mean_ncg + 1) * 8 * n
    return total + mean_ncg * vec_passes * 8 * P + 64 * n + 30 * 8 * P


def multi_chain_ceiling(k, ms_step, n_apply, op_ms, shared_bytes,
                        chain_bytes, batch_passes=1.):
    """What the batched formulation allows at best, to read `vs_k1` against: a
    batch step = the operator applications of ONE chain with the
    chain-independent bytes of a product (id stream, row ids, schedules, or
    the dense matrix) moved once and the per-chain bytes (vectors, slabs) k
    times, at the single chain's rate, plus k times what is left of a single
    chain's iteration (Polya-Gamma, scales, normals, vector kernels: per
    chain).  ceiling = k * ms_step / that.  batch_passes: how often the batch
    reads the shared bytes per application where one chain reads them once (2
    for dense designs: K chains' state does not fit a CU, so the single-pass
    operator becomes two passes)."""
    t_apply = n_apply * op_ms
    t_rest = max(ms_step - t_apply, 0.)
    grow = (batch_passes * shared_bytes + k * chain_bytes) / \
        (shared_bytes + chain_bytes)
    return k * ms_step / (t_apply * grow + k * t_rest)


def multi_chain_block(design, make_chain, state, widths, steps, warmup,
                      single_value, dense, ceiling_of=None):
    """k chains on one GPU through ONE pass over X per product (csrc/batch.hip)
    -- the reference's answer to "more chains" is more processes, each with
    its own passes (bayesbridge.py:109).  Every batch starts all its chains
    from the single chain's post-warm-up state (own seeds), runs `warmup`
    untimed and `steps` timed iterations and reports aggregate
    chain-iterations/s, the batched launches' bytes and their rate."""
    import numpy as np
    import torch
    from bayesbridge_amd import HipChainBatch
    coef, obs, ls, g, mean, square, navg = state
    out = {"what": "k chains per GPU sharing every pass over X (K-column "
                   "products); chain-iterations/s of the whole batch, same "
                   "start state and stationarity as the single-chain line; "
                   "`ceiling` = the speed-up the formulation allows (matrix "
                   "bytes once, per-chain bytes and non-CG work k times, at "
                   "the single chain's rates), `of_ceiling` = vs_k1 / ceiling",
           "k=1": {"chain_iters_per_sec": round(single_value, 2)}}
    for k in widths:
        chains = []
        for i in range(k):
            ch = make_chain(7000 + 13 * i)
            ch.set_state(coef, obs, ls, g)
            ch.set_summary(mean, square, navg)
            chains.append(ch)
        predicted = HipChainBatch.predicted_speedup(design, k)
        t0 = time.perf_counter()
        # (allow_slow: the line SHOWS the widths the library refuses by default)
        batch = HipChainBatch(chains, allow_slow=True)
        build_s = time.perf_counter() - t0
        batch.run_device(warmup)
        design.set_timing(True, every=8)
        design.reset_timing()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gs, lp, ncg, _ = batch.run_device(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        timing = design.get_timing()
        design.set_timing(False)
        assert np.all(np.isfinite(lp)) and np.all(gs > 0)
        dot_b, tdot_b = batch.launch_bytes
        grids = None
        if not dense and k in (2, 4):
            try:
                grids = {kk: vv["grid"] for kk, vv in
                         design.tiled_info(chains=k).items()}
            except Exception:      # noqa: BLE001 (mixed designs: other slots)
                grids = None
        entry = {"chain_iters_per_sec": round(k * steps / dt, 2),
                 "launch_grids": grids,
                 "vs_k1": round(k * steps / dt / single_value, 3),
                 "ms_per_batch_step": round(1e3 * dt / steps, 4),
                 "mean_n_cg_iter": [round(float(v), 1) for v in ncg.mean(1)],
                 "layout_build_s": round(build_s, 2),
                 # bbx_batch_predict: < 1 => bbx_batch_create refuses the width
                 "predicted_product_speedup": round(predicted, 3),
                 "refused_by_default": bool(predicted < 1.)}
        if ceiling_of is not None:
            ceil = ceiling_of(k)
            entry["ceiling"] = round(ceil, 3)
            entry["of_ceiling"] = round(k * steps / dt / single_value / ceil, 3)
        for name, nbytes in (("dot", dot_b), ("tdot", tdot_b)):
            cnt, ms = timing[name]
            if cnt > 0 and ms > 0:
                avg = ms / cnt
                entry[name] = {"avg_ms": round(avg, 5), "bytes": int(nbytes),
                               "gbs": round(nbytes / avg / 1e6, 1),
                               "frac": round(nbytes / avg / 1e6 / HBM_PEAK_GBS,
                                             4), "launches": cnt}
        out["k=%d" % k] = entry
        batch.close()
        for ch in chains:
            ch.close()
    return out


def committed_traffic(design, which, cfg):
    """HBM bytes per launch of the dominant kernel from the committed PMC
    passes (profiles/r0N_spmv_profile.json; FETCH_SIZE doubled per the gfx950
    correction + WRITE_SIZE), matched by launch grid.  NOT measured in this
    run: PMC collection needs its own rocprofv3 passes.  (bytes, source) or
    (None, None) when no profile of this workload/geometry is committed."""
    if cfg != "config3" or design.storage_format != "tiled":
        return None, None
    info = design.tiled_info()["X" if which == "dot" else "Xt"]
    n, P = design.shape
    rows = n if which == "dot" else P - 1
    n_wg = -(-rows // info["PR"]) * info["G"]
    for name in ("r06_spmv_profile.json", "r05_spmv_profile.json",
                 "r04_spmv_profile.json",
                 "r03_spmv_traffic.json",
                 "r02_spmv_profile.json", "r01_spmv_profile.json"):
        tag = name.split("_")[0]
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            prof = json.load(fh)
        entry = prof.get("hbm_traffic", {}).get("grid=%d" % n_wg)
        if entry:
            return int(entry["total_bytes"]), \
                "profiles/%s (separate rocprofv3 --pmc passes of the same " \
                "kernel and matrix; not collected in this run)" % name
    return None, None


def live_traffic(cfg, n_wg, reps=6, timeout=200):
    """HBM bytes per launch of the dominant kernel MEASURED in this run: two
    `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE: separate passes, the
    program itself after `--`, never combined with tracing) over
    scripts/bench_spmv.py -- the same generator, seed and layout builder, hence
    the same matrix, layout and launch grids as the chain above; the product
    kernels alone -- as CHILD processes of this one, after the timed region.
    bytes = 2 * FETCH_SIZE + WRITE_SIZE (KiB; MI355X_MICROARCH.md, HBM: on
    gfx950 FETCH_SIZE reports half the bytes of a 16-byte-per-lane stream).
    Returns (bytes, source) or (None, reason)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    # (this process itself under a profiler: its preloaded tool library would
    # be inherited by the child passes)
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ) \
            or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "bench.py runs under a profiler itself"
    child_env = {k: v for k, v in os.environ.items()
                 if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK",
                              "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    child_env["TMPDIR"] = "/tmp"
    got = {}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="bbx_pmc_", dir="/tmp")
        try:
            run = subprocess.run(
                [exe, "--pmc", counter, "--output-format", "csv", "-d", out,
                 "--", sys.executable,
                 os.path.join(ROOT, "scripts", "bench_spmv.py"), cfg, "tiled",
                 str(reps)],
                cwd="/tmp", env=child_env,
                capture_output=True, text=True, timeout=timeout)
            vals = []
            for path in glob.glob(os.path.join(out, "**",
                                               "*counter_collection.csv"),
                                  recursive=True):
                with open(path, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if "tiled_spmv" in row.get("Kernel_Name", "") \
                                and row.get("Counter_Name") == counter \
                                and int(row["Grid_Size"]) // 1024 == n_wg:
                            vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, "rocprofv3 --pmc %s: no launch of grid %d seen " \
                    "(rc %d: %s)" % (counter, n_wg, run.returncode,
                                     (run.stderr or "")[-200:])
            got[counter] = sum(vals) / len(vals)
        except Exception as exc:      # noqa: BLE001 (never costs the line)
            return None, "rocprofv3 --pmc %s failed: %s" % (counter, exc)
        finally:
            shutil.rmtree(out, ignore_errors=True)
    total = 1024. * (2. * got["FETCH_SIZE"] + got["WRITE_SIZE"])
    return int(total), (
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, two "
        "child passes over scripts/bench_spmv.py %s tiled (the product kernel "
        "alone on the same matrix and layout, %d launches of grid %d each; "
        "2 x FETCH_SIZE + WRITE_SIZE; %.0f s)"
        % (cfg, len(vals), n_wg, time.perf_counter() - t0))


def main():
    t_proc = time.perf_counter()
    args = parse_args()

    def progress(msg):
        # milestones of a rank on stderr (multi-rank runs and BENCH_PROGRESS=1):
        # where the set-up time of an N-rank launch goes
        if os.environ.get("WORLD_SIZE") or os.environ.get("BENCH_PROGRESS"):
            sys.stderr.write("[bench rank %s +%.1fs] %s\n" % (
                os.environ.get("RANK", "0"), time.perf_counter() - t_proc, msg))
            sys.stderr.flush()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        # self-launch: nothing here has imported torch or touched HIP yet
        from bayesbridge_amd.chains import launch_ranks
        assert "torch" not in sys.modules
        sys.exit(launch_ranks(args.gpus, [os.path.abspath(__file__)]
                              + sys.argv[1:]))
    if env_world is not None and int(env_world) != args.gpus:
        sys.stderr.write("bench.py: --gpus %d does not match WORLD_SIZE=%s "
                         "(launch with --nproc-per-node %d, or let bench.py "
                         "launch the ranks itself by unsetting WORLD_SIZE)\n"
                         % (args.gpus, env_world, args.gpus))
        sys.exit(2)
    # stdout carries exactly ONE line (the JSON result of rank 0): libraries
    # that write to file descriptor 1 (gloo's connection notes, rocm tools)
    # are sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    from bayesbridge_amd import (HipDenseDesignMatrix, HipGibbsChain,
                                 HipSparseDesignMatrix, _lib, chains)
    # under a launcher a process group exists even for one rank, so that
    # `torch.distributed.run --nproc-per-node 1 bench.py` takes the N-rank path
    # (RCCL initialisation, gather, MAX all-reduce, barrier) on a 1-GPU box
    rank, world, local_rank = chains.init_process_group_from_env(
        single_rank_group=env_world is not None)
    n_dev = torch.cuda.device_count()
    progress("process group ready (world %d)" % world)
    backend = None
    grouped = world > 1 or env_world is not None
    if grouped:
        import torch.distributed as dist
        backend = dist.get_backend()
        if n_dev >= world:
            # one rank per GPU over RCCL (backend 'nccl' IS RCCL on ROCm)
            assert backend == "nccl", backend
            assert local_rank < n_dev
    dev_index = local_rank % max(n_dev, 1)   # == local_rank on a full node
    torch.cuda.set_device(dev_index)
    device = "cuda:%d" % dev_index
    if grouped and n_dev >= world:
        # every rank really sits on its own device
        ids = [None] * world
        import torch.distributed as dist
        dist.all_gather_object(ids, dev_index)
        assert len(set(ids)) == world, ids

    dense = CONFIGS[args.config][2] is None
    if args.cpu_baseline_iters is None:
        args.cpu_baseline_iters = 2 if dense else 5
    solo = rank == 0 and world == 1 and env_world is None
    x_host = None
    unit = math.gamma(2 / ALPHA) / math.gamma(1 / ALPHA)   # prior.py:163-167
    seed_k = chains.chain_seed(args.seed, rank)
    if dense:
        with chains.setup_turn():      # (one rank at a time on a shared GPU)
            prob = build_dense_problem(torch, args.config, args.seed, device)
        n, p, nnz = prob["n"], prob["p"], prob["nnz"]
        torch.cuda.synchronize()
        design = HipDenseDesignMatrix.from_device_array(
            n, p, prob["X"].data_ptr(), prob["offset"].data_ptr(),
            add_intercept=True, device=dev_index, in_dtype='float32',
            storage_dtype=args.dense_storage)
        if solo and args.cpu_baseline_iters > 0:
            # the CPU leg runs on the same matrix (f32 here, widened there)
            x_host = prob["X"].cpu().numpy()
        del prob["X"]
        outcome = prob["y"].cpu().numpy()
        def make_chain(seed):
            return HipGibbsChain(design, 'linear', outcome,
                                 bridge_exponent=ALPHA, slab_size=SLAB,
                                 seed=seed)
        chain = make_chain(seed_k)
        intercept0 = outcome.mean()
    else:
        with chains.setup_turn():      # (one rank at a time on a shared GPU)
            prob = build_problem(torch, args.config, args.seed, device)
            torch.cuda.synchronize()
        n, p, nnz = prob["n"], prob["p"], prob["nnz"]
        progress("design generated in HBM (nnz %d)" % nnz)
        design = HipSparseDesignMatrix.from_device_csr(
            n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(),
            None, prob["offset"].data_ptr(), add_intercept=True,
            device=dev_index, storage=args.storage)
        if args.cg_fold is not None:
            design.set_cg_fold(bool(args.cg_fold))
        # chain: prior and init of the reference demo (demo.ipynb cells 7, 9)
        n_success = prob["n_success"].cpu().numpy()
        def make_chain(seed):
            return HipGibbsChain(design, 'logit', n_success,
                                 bridge_exponent=ALPHA, slab_size=SLAB,
                                 seed=seed)
        chain = make_chain(seed_k)
        ph = n_success.mean()
        intercept0 = math.log(ph / (1 - ph))             # intercept MLE
    P = p + 1
    coef0 = np.zeros(P)
    coef0[0] = intercept0
    # init global_scale=.01 in the user parametrisation (prior.py:129-141)
    chain.set_state(coef0, None, np.ones(P - 1) * unit, .01 / unit)
    chain.init_obs_prec()

    torch.cuda.synchronize()
    startup_s = time.perf_counter() - t_proc   # import, generate, build layout
    progress("layouts built, chain ready (%d builder threads)"
             % _lib.builder_threads())
    K, W, B = args.steps, args.warmup, args.burnin
    if B is None:
        B = 100 if dense else 300
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    ncg_b = chain.run_device(B)[2] if B > 0 else np.zeros(0)
    burnin_ms = 1e3 * (time.perf_counter() - t_b) / max(B, 1)
    progress("burn-in done (%d iterations, %.1f ms each)" % (B, burnin_ms))
    # Everything host-side that the timed region needs happens BEFORE the W
    # warm-up steps (state for the CPU leg, sample buffer, event pool, the
    # process group's first gather), so that the GPU runs Gibbs iterations
    # right up to t0: a gap here lets the clocks fall and block 0 paid for the
    # ramp (LABNOTES R5.1).
    widths = args.multi_chain
    if widths is None:
        widths = "4,8,16,32" if dense else "2,4"
    widths = [int(v) for v in widths.split(",") if int(v) > 1]
    want_state = solo and (args.cpu_baseline_iters > 0 or bool(widths))
    d_buf = torch.empty((max(K, W, 1), P), dtype=torch.float64, device=device)
    d_coef = d_buf[:max(K, 1)]          # (the warm-up may keep more samples)
    if grouped:
        # the first gather of a process group sets up the point-to-point
        # connections (RCCL does that lazily, 100s of ms)
        chains.gather_chain_samples(d_coef, dst=0)
    # kernel stamps / brackets on one launch in --timing-every (64: a stamped
    # launch costs ~8 us, one in 16 was 1 % of `value` -- LABNOTES R5.11; timing
    # every launch costs ~10 % of the iteration); switched on before the
    # warm-up so that the event pool exists and the warm-up runs the same code
    timing_on = args.timing_blocks != "none"
    if timing_on:
        design.set_timing(True, every=args.timing_every or 16)
    ncg_w = chain.run_device(W, d_coef_ptr=d_buf.data_ptr())[2] \
        if W > 0 else np.zeros(0)
    if timing_on:
        if not args.timing_every:
            # ~24 stamped launches per kernel and K-step block
            per_block = K * (float(np.mean(ncg_w)) if len(ncg_w) else 30.)
            args.timing_every = int(min(64, max(8, per_block // 24)))
            design.set_timing(True, every=args.timing_every)
        design.reset_timing()
    lib = _lib.load()
    has_stats = hasattr(design, "cg_stats")      # (A/B against an older build)
    if has_stats:
        design.cg_stats(reset=True)
    chains.barrier()
    torch.cuda.synchronize()
    launches0 = lib.bbx_launch_count() if has_stats else 0
    cpu0 = time.process_time()
    t0 = time.perf_counter()
    gs, lp, ncg, _ = chain.run_device(K, d_coef_ptr=d_coef.data_ptr())
    t_run = time.perf_counter()
    own_cpu = time.process_time() - cpu0
    launches = (lib.bbx_launch_count() - launches0) if has_stats else None
    cg_solves, cg_empty, cg_naps = design.cg_stats() if has_stats else (0, 0, 0)
    gathered = chains.gather_chain_samples(d_coef, dst=0)
    if grouped:
        torch.cuda.synchronize()
    t_gather = time.perf_counter()
    chains.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    own_elapsed, own_run, own_gather = elapsed, t_run - t0, t_gather - t_run
    # the host side of this rank's K iterations: CPU seconds this process
    # burnt (enqueueing + polling the CG loop's progress word), kernel
    # launches, and the launches enqueued past a solve's stopping iteration
    host_side = dict(
        cpu_s_per_step=round(own_cpu / max(K, 1), 6),
        cpu_busy_frac=round(own_cpu / own_run, 3) if own_run > 0 else None,
        launches_per_step=round(launches / max(K, 1), 1)
        if launches is not None else None,
        launches_per_sec=round(launches / own_run, 0)
        if launches is not None and own_run > 0 else None,
        empty_launches_per_draw=round(cg_empty / max(cg_solves, 1), 2)
        if has_stats else None,
        # > 0: the host slept between stop tests (fewer than three cores
        # per rank, or BBX_CG_SLEEP=1) instead of polling
        naps_per_draw=round(cg_naps / max(cg_solves, 1), 1)
        if has_stats else None)
    elapsed = chains.max_over_ranks(elapsed)
    progress("timed region done (%.3f s)" % elapsed)
    timing = design.get_timing() if timing_on else {
        "dot": (0, 0.), "tdot": (0, 0.), "operator": (0, 0.)}
    if args.timing_blocks == "first":
        design.set_timing(False)
    # state after the timed block (for the CPU baselines and the batches): the
    # chain is stationary, any post-burn-in state serves
    state = None
    if want_state:
        coef, obs, ls, g = chain.get_state()
        mean, square, navg = chain.get_summary()
        state = (coef, obs, ls, g, mean, square, navg)

    # `repeat`: the same K-step block four more times, each bracketed like the
    # timed region (barrier + synchronize, max over ranks); value/steps/
    # ms_per_step above are the FIRST block and stay what they were.  Gives the
    # line a spread: box-to-box and block-to-block noise is +-1.5 %.
    block_values = [world * K / elapsed]
    block_ncg = [float(ncg.mean())]
    for _ in range(max(args.repeat - 1, 0)):
        if args.timing_blocks == "all":
            design.reset_timing()      # (hands the event pairs back to the pool)
        chains.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ncg_r = chain.run_device(K, d_coef_ptr=d_coef.data_ptr())[2]
        chains.gather_chain_samples(d_coef, dst=0)
        chains.barrier()
        torch.cuda.synchronize()
        dt_r = chains.max_over_ranks(time.perf_counter() - t0)
        block_values.append(world * K / dt_r)
        block_ncg.append(float(ncg_r.mean()))
    if args.timing_blocks == "all":
        design.set_timing(False)
    # set-up cost of EVERY rank (eight generators and eight layout builders
    # run side by side on one host at config 5)
    rss_mb = int(__import__("resource").getrusage(
        __import__("resource").RUSAGE_SELF).ru_maxrss / 1024)
    per_rank = None
    if grouped:
        import torch.distributed as dist
        box = [None] * world
        # what each rank saw of the timed region BEFORE the MAX all-reduce: if
        # an N-GPU line is not N x the 1-GPU line, this says which rank, and
        # whether it was the chain, the gather or the set-up
        mine = dict(
            startup_s=round(startup_s, 1), peak_host_rss_mb=rss_mb,
            builder_threads=_lib.builder_threads(),
            timed_s=round(own_elapsed, 4), run_s=round(own_run, 4),
            gather_ms=round(1e3 * own_gather, 3),
            iters_per_sec=round(K / own_run, 2) if own_run > 0 else None,
            mean_n_cg_iter=round(float(ncg.mean()), 2),
            burnin_ms_per_step=round(burnin_ms, 4),
            host_cpu_s_per_step=host_side["cpu_s_per_step"],
            host_cpu_busy_frac=host_side["cpu_busy_frac"],
            launches_per_sec=host_side["launches_per_sec"],
            empty_launches_per_draw=host_side["empty_launches_per_draw"],
            device_index=dev_index,
            device_name=torch.cuda.get_device_name(dev_index),
            pid=os.getpid())
        dist.all_gather_object(box, mine)
        per_rank = {key: [b[key] for b in box] for key in mine}
        per_rank["what"] = (
            "one entry per rank: timed_s = this rank's barrier-to-barrier "
            "seconds before the MAX over ranks, run_s = its K Gibbs "
            "iterations alone, gather_ms = its share of the one gather of "
            "the kept samples, iters_per_sec = K / run_s; host_cpu_s_per_step "
            "= CPU seconds of this process per Gibbs iteration "
            "(time.process_time), host_cpu_busy_frac = that / run_s (1.0 = "
            "one core: the host polls the CG loop's progress word), "
            "launches_per_sec = kernel launches of this rank per second")

    if rank == 0:
        assert gathered is not None and gathered.shape[0] == world
        assert bool(torch.isfinite(gathered).all())
        assert np.all(np.isfinite(lp)) and np.all(gs > 0)
        ms_step = 1e3 * elapsed / K
        mean_ncg = float(ncg.mean())
        dot_tb, tdot_tb = design.timed_bytes      # what the stamps cover
        dot_wb, tdot_wb = design.matvec_bytes     # whole products
        fused_b = design.fused_operator_bytes if dense else 0
        per = {}
        for name, b in (("dot", dot_tb), ("tdot", tdot_tb)):
            cnt, ms = timing[name]
            avg_ms = ms / max(cnt, 1)
            per[name] = dict(launches=cnt, avg_ms=avg_ms, bytes=b,
                             gbs=b / avg_ms / 1e6 if avg_ms > 0 else 0.)
        if dense and fused_b:
            # inside the CG loop family 0 is the single-pass operator kernel
            per["dot"]["bytes"] = fused_b
            per["dot"]["gbs"] = fused_b / per["dot"]["avg_ms"] / 1e6 \
                if per["dot"]["avg_ms"] > 0 else 0.
        dom = "dot" if timing["dot"][1] >= timing["tdot"][1] else "tdot"
        ach = per[dom]["gbs"]
        # the same launch on the bytes a reader cannot do without: no padding of
        # the 16-byte steps, no schedules (bbx_design_useful_bytes)
        (use_dot, use_tdot), (pad_dot, pad_tdot) = design.useful_bytes
        if dense and fused_b:
            use_dot = fused_b
        useful_b = use_dot if dom == "dot" else use_tdot
        useful_gbs = useful_b / per[dom]["avg_ms"] / 1e6 \
            if per[dom]["avg_ms"] > 0 else 0.
        traffic, traffic_src = committed_traffic(design, dom, args.config)
        traffic_committed = traffic
        if solo and args.live_traffic and not dense \
                and design.storage_format == "tiled" \
                and args.config in ("config2", "config3") \
                and args.seed == 111 and args.storage in ("auto", "tiled"):
            info_dom = design.tiled_info()["X" if dom == "dot" else "Xt"]
            live, live_src = live_traffic(args.config, int(info_dom["grid"]))
            if live is not None:
                traffic, traffic_src = live, live_src
            else:
                traffic_src = "%s; live measurement skipped: %s" % (
                    traffic_src, live_src)
        # whole operator application (dot + Tdot + epilogue kernel)
        op_cnt, op_ms = timing["operator"]
        op_avg = op_ms / max(op_cnt, 1)
        if dense and fused_b:
            ld = -(-P // 8) * 8       # + the epilogue's slab read and output
            op_bytes = fused_b + 8 * 256 * ld + 8 * P
        else:
            op_bytes = dot_wb + tdot_wb
        op_gbs = op_bytes / op_avg / 1e6 if op_avg > 0 else 0.
        cg_launches = design.cg_launches
        iter_bytes = iteration_bytes(
            mean_ncg, op_bytes, dot_wb, tdot_wb, n, P, bool(dense and fused_b),
            vec_passes=17 if (cg_launches == 3 and not dense) else 15)
        iter_gbs = iter_bytes / ms_step / 1e6
        # measured on this box, same size as one launch's algorithmic bytes
        # and at 2 GB: what a plain streaming kernel reaches
        probe_small = _lib.hbm_probe(int(per[dom]["bytes"]), 50, dev_index)
        probe_large = _lib.hbm_probe(2 << 30, 10, dev_index)
        # Workloads whose two orientations fit the 256 MiB Infinity Cache next
        # to each other are re-read from the die, launch after launch: the HBM
        # peak is not their ceiling.  The same-size read probe above re-reads
        # its buffer 50 times, i.e. measures exactly that on-die rate.
        cache_bound = None
        if not dense and (dot_wb + tdot_wb) <= (256 << 20):
            cache_bound = dict(
                resident_bytes=int(dot_wb + tdot_wb),
                ceiling="plain read stream of the launch's bytes, re-read "
                        "50 times on this box (stream_probe_gbs."
                        "read_same_bytes): served by L2 / Infinity Cache",
                ceiling_gbs=round(probe_small[0], 1),
                frac=round(ach / probe_small[0], 4) if probe_small[0] > 0
                else None)
        kernel_name = ("operator X^T(Omega(X v)), one pass (dense %s)"
                       % ("f32" if args.dense_storage == "float32" else "f64")
                       if dense and fused_b and dom == "dot"
                       else dom + " (" + design.storage_format + ")")
        roofline = dict(
            bound="hbm", achieved=round(ach, 1), peak=HBM_PEAK_GBS,
            unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=traffic,
            traffic_source=traffic_src,
            traffic_committed_profile=traffic_committed,
            kernel=kernel_name,
            avg_launch_ms=round(per[dom]["avg_ms"], 5),
            algorithmic_bytes_per_launch=per[dom]["bytes"],
            useful_bytes_per_launch=int(useful_b),
            useful_frac=round(useful_gbs / HBM_PEAK_GBS, 4),
            useful_what="the launch's bytes without the padding of the id "
                        "steps and without the schedules: stored entries x "
                        "index bytes + row ids + vector in + output",
            timing=("kernel begin/end stamps (hipExtLaunchKernelGGL events) on "
                    "the launching stream, one launch in %d, inside the timed "
                    "region" if design.storage_format == "tiled" else
                    "hipEventRecord bracket on the launching stream, one launch "
                    "in %d, inside the timed region") % args.timing_every,
            operator_frac=round(op_gbs / HBM_PEAK_GBS, 4),
            operator=dict(avg_ms=round(op_avg, 5), bytes=int(op_bytes),
                          gbs=round(op_gbs, 1), launches=op_cnt,
                          what="one application of the CG operator: dot + "
                               "Tdot + epilogue kernel, event bracket"),
            iteration_frac=round(iter_gbs / HBM_PEAK_GBS, 4),
            iteration=dict(bytes=int(iter_bytes), gbs=round(iter_gbs, 1),
                           what="algorithmic bytes of one whole Gibbs "
                                "iteration / ms_per_step"),
            cache_bound=cache_bound,
            stream_probe_gbs={
                "read_same_bytes": round(probe_small[0], 1),
                "copy_same_bytes": round(probe_small[1], 1),
                "read_2GiB": round(probe_large[0], 1),
                "copy_2GiB": round(probe_large[1], 1)},
            other={k: dict(avg_ms=round(v["avg_ms"], 5),
                           gbs=round(v["gbs"], 1), launches=v["launches"],
                           bytes=v["bytes"])
                   for k, v in per.items()})
        line = {
            "metric": "Gibbs iters/sec (cg sampler)",
            "value": round(world * K / elapsed, 4),
            "unit": "Gibbs iters/sec",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": (
                    "%s: linear, dense N(0,1) %dx%d stored %s (torch Philox "
                    "seed %d), centred + intercept, one independent chain "
                    "per GPU, seeds %d+rank"
                    % (args.config, n, p,
                       "f32" if args.dense_storage == "float32" else "f64",
                       args.seed, args.seed)) if dense else (
                    "%s: logit, sparse binary CSR %dx%d nnz=%d "
                    "(simulate_data.py distribution, f=%g), one "
                    "independent chain per GPU, seeds %d+rank"
                    % (args.config, n, p, nnz, CONFIGS[args.config][2],
                       args.seed)),
                "storage": design.storage_format,
                "cg_launches_per_iteration": cg_launches,
                # workgroups per launch of the two product kernels (what the
                # rocprofv3 traces under profiles/ are keyed by)
                "launch_grids": {k: v["grid"] for k, v in
                                 design.tiled_info().items()}
                if design.storage_format == "tiled" else None,
                # how the value-free ids are stored: four 16-bit ids or one
                # group of five entries per eight bytes
                "id_format": {k: ("groups of 5 / 8 B" if v["packed"]
                                  else "4 ids / 8 B") for k, v in
                              design.tiled_info().items()}
                if design.storage_format == "tiled" else None,
                # share of the id stream that is padding (steps are 16 bytes
                # per lane: a row segment rounds up to whole groups / quads, a
                # slice to its longest row)
                "id_padding": {"X": round(pad_dot, 4), "Xt": round(pad_tdot, 4)}
                if design.storage_format == "tiled" else None,
                "init": "coef=0 + intercept MLE, global_scale=.01, then %d "
                        "untimed burn-in iterations (in place of the "
                        "reference's L-BFGS mode search)" % B,
                "mean_n_cg_iter": round(mean_ncg, 2),
                "mean_n_cg_iter_warmup": round(float(ncg_w.mean()), 2)
                if W > 0 else None,
                "mean_n_cg_iter_burnin": round(float(ncg_b.mean()), 2)
                if B > 0 else None,
                # the chain's transient, for comparison with `ms_per_step`
                # (same code, more CG iterations per draw)
                "burnin_ms_per_step": round(burnin_ms, 4) if B > 0 else None,
                "first_50_n_cg_iter": round(float(ncg_b[:50].mean()), 2)
                if B >= 50 else None,
                # the timed region's counts (what the solve's look-ahead for
                # the stop flag has to predict)
                "n_cg_iter_timed": [int(v) for v in ncg[:64]],
                # this rank's host side of the timed K iterations
                "host": host_side,
                # set-up cost per rank (an 8-rank launch runs 8 generators and 8
                # host-side layout builders side by side): seconds from process
                # start to a ready chain and this rank's peak host RSS
                "startup_s": round(startup_s, 1),
                "peak_host_rss_mb": rss_mb,
                # host threads of the layout builder (affinity, cgroup quota,
                # LOCAL_WORLD_SIZE; csrc/tiled_layout.cpp builder_threads)
                "builder_threads": _lib.builder_threads(),
                "per_rank": per_rank,
                "parallelism": "chains=%d" % world,
                "devices": min(world, n_dev),
                "backend": backend,
                "rccl_ranks": world if backend == "nccl" else 0,
            },
            "roofline": roofline,
            "repeat": {
                "what": "the timed K-step block run %d times back to back "
                        "(block 0 is `value`); Gibbs iters/sec of each and "
                        "their median; us_per_cg_iter = a block's wall time per "
                        "CG iteration of one chain (n_cg differs from block to "
                        "block, the cost of an iteration does not)"
                        % len(block_values),
                "values": [round(v, 2) for v in block_values],
                "median": round(float(np.median(block_values)), 2),
                "spread_pct": round(100. * (max(block_values)
                                            - min(block_values))
                                    / float(np.median(block_values)), 2),
                "mean_n_cg_iter": [round(v, 2) for v in block_ncg],
                # a block's time follows its CG iterations: this is what is
                # flat from block to block (LABNOTES R5.1)
                "us_per_cg_iter": [round(1e6 * world / v / max(c, 1e-9), 2)
                                   for v, c in zip(block_values, block_ncg)],
                "kernel_stamps": args.timing_blocks},
        }
        if state is not None and widths:
            # (an extra block beside the headline: it must never cost the line)
            try:
                # bytes of one operator application that do not depend on the
                # chain: everything but the vectors and slabs
                passes = 1.
                if dense:
                    el = 4 if args.dense_storage == "float32" else 8
                    mat_b = float(n * (-(-P // 8) * 8) * el)
                    shared_b = mat_b * (1 if fused_b else 2)
                    passes = 2. if fused_b else 1.
                else:
                    shared_b = float(design.storage_bytes)
                chain_b = max(float(op_bytes) - shared_b, 0.)
                n_apply = mean_ncg + 1.      # + the initial residual's products

                def ceiling_of(k):
                    return multi_chain_ceiling(k, ms_step, n_apply, op_avg,
                                               shared_b, chain_b, passes)
                line["multi_chain"] = multi_chain_block(
                    design, make_chain, state, widths, args.multi_chain_steps,
                    5, line["value"], dense, ceiling_of)
            except Exception as exc:     # noqa: BLE001
                line["multi_chain"] = {"error": "%s: %s" % (type(exc).__name__,
                                                            exc)}
        if state is not None and args.cpu_baseline_iters > 0 and not dense:
            port, omp = cpu_baselines(prob, state, args.cpu_baseline_iters,
                                      args.seed)
            line["cpu_baseline"] = port
            line["cpu_baseline_omp"] = omp
        elif state is not None and args.cpu_baseline_iters > 0:
            line["cpu_baseline"] = cpu_baseline_dense(
                x_host, prob["offset"].cpu().numpy(), outcome, state,
                args.cpu_baseline_iters, args.seed)
        else:
            line["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    chain.close()
    if grouped:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
