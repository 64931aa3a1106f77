#!/usr/bin/env python3
"""Does bench.py's `cpu_baseline` (kind "port": the oracle chain on SciPy's
CSR products and scipy.sparse.linalg.cg) track the REFERENCE's wall-clock?

BUILD CONTAINER ONLY: imports the reference from /root/reference through
tests/golden/ref_import.py (scratch copy under /tmp, nothing of it enters this
repo or travels to the GPU box).  SURVEY 8(d) asks for "identical n_cg_iter,
coef within 1e-10, wall-clock within +-10 % on configs 1-2"; VERDICT r04 #5
measured 0.87x (the port faster) and asked for the record.

For BASELINE configs 1 (linear, dense 2000 x 500) and 2 (logit, sparse binary
100 000 x 10 000, nnz 1.02e7, full size), same design, outcome, prior, init and
seed: N Gibbs iterations of
    reference   bayesbridge.BayesBridge.gibbs(coef_sampler_type='cg')
    port        oracle.OracleGibbs(use_scipy_cg=True).gibbs
each timed as a whole (mode search included, as gibbs() runs it) and, for the
port, split by part; the chains are run --repeat times alternately and the
minimum is kept (the container's 8 cores are shared).  Writes a text report.

    python scripts/validate_cpu_port.py [--iters 10] [--repeat 3] \
        [--out profiles/r05_cpu_port_validation.txt]
"""
import argparse
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_import  # noqa: E402

warnings.simplefilter("ignore")


def problems(refsim):
    beta_head = np.zeros(15)
    beta_head[:5], beta_head[5:10], beta_head[10:15] = 1.5, 1., .5
    # config 1: simulate_design(2000, 500, format_='dense', seed=111), linear
    X1 = refsim.simulate_design(2000, 500, binary_frac=0., format_='dense',
                                seed=111)
    b1 = np.zeros(500)
    b1[:15] = beta_head
    y1 = refsim.simulate_outcome(X1, b1, 'linear', seed=1)
    yield ("config1: linear, dense 2000x500", 'linear', X1, y1, .1)
    # config 2: the literal generator (8 GB transient, ~60 s)
    X2 = refsim.simulate_design(100000, 10000, binary_frac=1.,
                                binary_pred_freq=.01, format_='sparse',
                                seed=111).tocsr()
    X2.sort_indices()
    b2 = np.zeros(10000)
    b2[:15] = beta_head
    y2 = refsim.simulate_outcome(X2, b2, 'logit', seed=1)
    yield ("config2: logit, sparse binary 100000x10000 nnz=%d" % X2.nnz,
           'logit', X2, y2, .01)


def run_reference(bb, family, X, y, gscale, n_iter, seed):
    model = bb.RegressionModel(y, X.copy() if family == 'linear' else X,
                               family)
    prior = bb.RegressionCoefPrior(bridge_exponent=.5,
                                   regularizing_slab_size=2.)
    bridge = bb.BayesBridge(model, prior)
    t0 = time.perf_counter()
    samples, info = bridge.gibbs(n_iter, 0, init={'global_scale': gscale},
                                 coef_sampler_type='cg', seed=seed)
    dt = time.perf_counter() - t0
    return dt, samples['coef'], info['_reg_coef_sampling_info']['n_cg_iter']


def run_port(family, X, y, gscale, n_iter, seed):
    from oracle.gibbs import OracleGibbs
    chain = OracleGibbs(y, X.copy() if family == 'linear' else X, family,
                        bridge_exponent=.5, regularizing_slab_size=2.,
                        use_scipy_cg=True)
    # per-part seconds of the port (wrappers around the bound methods)
    parts = {}

    def timed(name):
        fn = getattr(chain, name)

        def wrapper(*a, **k):
            t = time.perf_counter()
            out = fn(*a, **k)
            parts[name] = parts.get(name, 0.) + time.perf_counter() - t
            return out
        setattr(chain, name, wrapper)
    for name in ("draw_coef", "draw_obs_prec", "draw_gscale", "draw_lscale",
                 "logp"):
        timed(name)
    t0 = time.perf_counter()
    out = chain.gibbs(n_iter, seed=seed, init={'global_scale': gscale})
    dt = time.perf_counter() - t0
    parts["mode search + rest"] = dt - sum(parts.values())
    return dt, out['coef'], out['n_cg_iter'], parts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--seed", type=int, default=111)
    ap.add_argument("--out", default=os.path.join(
        ROOT, "profiles", "r05_cpu_port_validation.txt"))
    args = ap.parse_args()
    bb, refsim = ref_import.import_reference()
    import scipy
    lines = [
        "CPU `port` baseline of bench.py against the reference itself "
        "(scripts/validate_cpu_port.py;",
        "build container: %d CPUs, NumPy %s, SciPy %s; reference imported "
        "from /root/reference, not shipped)." % (
            os.cpu_count(), np.__version__, scipy.__version__),
        "%d Gibbs iterations incl. the initial mode search, seed %d, best of "
        "%d alternating runs." % (args.iters, args.seed, args.repeat), ""]
    for name, family, X, y, gscale in problems(refsim):
        t_ref, t_port = [], []
        for _ in range(args.repeat):
            dt, coef_r, ncg_r = run_reference(bb, family, X, y, gscale,
                                              args.iters, args.seed)
            t_ref.append(dt)
            dt, coef_p, ncg_p, parts = run_port(family, X, y, gscale,
                                                args.iters, args.seed)
            t_port.append(dt)
        err = float(np.abs(coef_r - coef_p).max())
        lines += [
            name,
            "  reference  %.3f s (runs: %s)" % (
                min(t_ref), " ".join("%.3f" % v for v in t_ref)),
            "  port       %.3f s (runs: %s)" % (
                min(t_port), " ".join("%.3f" % v for v in t_port)),
            "  port / reference wall-clock = %.3f" % (min(t_port) / min(t_ref)),
            "  n_cg_iter reference %s" % [int(v) for v in ncg_r],
            "  n_cg_iter port      %s  (sum %d vs %d)" % (
                [int(v) for v in ncg_p], int(np.sum(ncg_p)),
                int(np.sum(ncg_r))),
            "  max |coef_reference - coef_port| over all %d samples = %.3e "
            "(bitwise equal: %s)" % (args.iters, err,
                                     bool(np.array_equal(coef_r, coef_p))),
            "  port seconds by part (last run): " + ", ".join(
                "%s %.3f" % (k, v) for k, v in sorted(parts.items())),
            ""]
        print("\n".join(lines[-9:]))
        sys.stdout.flush()
    with open(args.out, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("written:", args.out)


if __name__ == "__main__":
    main()
