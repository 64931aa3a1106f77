"""A config-3 device chain for the HBM-traffic pin of bench.py's byte model:
`burnin` untimed iterations, then `iters` more, nothing else.  Run it under
`rocprofv3 --pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`;
scripts/summarize_iteration_traffic.py then sums the counters over the LAST
`iters` Gibbs iterations (from the iters-th last chain_prior_kernel on) and
writes profiles/r03_iteration_traffic.json.
Usage: python scripts/iteration_traffic.py [iters] [burnin] [out.json]"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (scripts/fold_phase_timers.sh points this at a private, instrumented build)
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import torch

from bayesbridge_amd import HipGibbsChain, HipSparseDesignMatrix
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
burnin = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out = sys.argv[3] if len(sys.argv) > 3 else None
prob = bench.build_problem(torch, "config3", 111, "cuda:0")
n, p, nnz = prob["n"], prob["p"], prob["nnz"]
torch.cuda.synchronize()
design = HipSparseDesignMatrix.from_device_csr(
    n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(), None,
    prob["offset"].data_ptr(), add_intercept=True, device=0, storage="auto")
n_success = prob["n_success"].cpu().numpy()
chain = HipGibbsChain(design, 'logit', n_success, bridge_exponent=bench.ALPHA,
                      slab_size=bench.SLAB, seed=111)
unit = math.gamma(2 / bench.ALPHA) / math.gamma(1 / bench.ALPHA)
P = p + 1
coef0 = np.zeros(P)
ph = n_success.mean()
coef0[0] = math.log(ph / (1 - ph))
chain.set_state(coef0, None, np.ones(P - 1) * unit, .01 / unit)
chain.init_obs_prec()
chain.run_device(burnin)
gs, lp, ncg, _ = chain.run_device(iters)
dot_wb, tdot_wb = design.matvec_bytes
info = dict(iters=iters, burnin=burnin, n=n, P=P, nnz=nnz,
            n_cg_iter=[int(v) for v in ncg], dot_bytes=int(dot_wb),
            tdot_bytes=int(tdot_wb),
            cg_launches=design.cg_launches,
            model_bytes_per_iteration=float(bench.iteration_bytes(
                float(ncg.mean()), dot_wb + tdot_wb, dot_wb, tdot_wb, n, P,
                vec_passes=17 if design.cg_launches == 3 else 15)))
print("ITERATION_TRAFFIC " + json.dumps(info))
if out:
    with open(out, "w") as fh:
        json.dump(info, fh)
