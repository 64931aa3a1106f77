"""n_cg per Gibbs iteration along a long device chain at a bench config: how
long until the CG warm start / preconditioner summaries are stationary?
Usage: python scripts/ncg_trajectory.py [config3] [n_iter]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np
import torch
import bench
from bayesbridge_amd import HipGibbsChain, HipSparseDesignMatrix

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 600
prob = bench.build_problem(torch, cfg, 111, "cuda:0")
torch.cuda.synchronize()
n, p, nnz = prob["n"], prob["p"], prob["nnz"]
design = HipSparseDesignMatrix.from_device_csr(
    n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(), None,
    prob["offset"].data_ptr(), add_intercept=True, device=0)
ns = prob["n_success"].cpu().numpy()
chain = HipGibbsChain(design, 'logit', ns, bridge_exponent=.5, slab_size=2.,
                      seed=111)
unit = math.gamma(4.) / math.gamma(2.)
coef0 = np.zeros(p + 1)
ph = ns.mean()
coef0[0] = math.log(ph / (1 - ph))
chain.set_state(coef0, None, np.ones(p) * unit, .01 / unit)
chain.init_obs_prec()
t0 = time.time()
gs, lp, ncg, _ = chain.run_device(n_iter)
dt = time.time() - t0
print("%s: %d iterations in %.2f s (%.1f it/s overall)" % (cfg, n_iter, dt, n_iter / dt))
for lo in range(0, n_iter, 50):
    hi = min(lo + 50, n_iter)
    print("  iterations %4d-%4d: mean n_cg %.1f  tau %.3g  logp %.6g" % (
        lo, hi, ncg[lo:hi].mean(), gs[lo:hi].mean(), lp[lo:hi].mean()))
