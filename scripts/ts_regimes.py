"""dev_ts_kernel time per regime at the chain's problem size (p = 50 000) and
the regime mix of a real chain state.  Run under rocprofv3 --kernel-trace."""
import ctypes
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np
from bayesbridge_amd import _lib

lib = _lib.load()
n = 50000
out = np.empty(n)
a = .25
for tp in (.1, 1., 1.9, 2.1, 4., 16., 100.):
    tilt = np.full(n, tp ** (1 / a))
    for rep in range(3):
        _lib.check(lib.bbx_device_tilted_stable(
            0, 1 + rep, n, a, tilt.ctypes.data_as(ctypes.c_void_p),
            out.ctypes.data_as(ctypes.c_void_p)))
if len(sys.argv) > 1:
    import torch
    import bench
    from bayesbridge_amd import HipGibbsChain, HipSparseDesignMatrix
    prob = bench.build_problem(torch, "config3", 111, "cuda:0")
    torch.cuda.synchronize()
    nn, p, nnz = prob["n"], prob["p"], prob["nnz"]
    design = HipSparseDesignMatrix.from_device_csr(
        nn, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(),
        None, prob["offset"].data_ptr(), add_intercept=True, device=0)
    ns = prob["n_success"].cpu().numpy()
    chain = HipGibbsChain(design, 'logit', ns, bridge_exponent=.5,
                          slab_size=2., seed=111)
    unit = math.gamma(4.) / math.gamma(2.)
    coef0 = np.zeros(p + 1)
    coef0[0] = math.log(ns.mean() / (1 - ns.mean()))
    chain.set_state(coef0, None, np.ones(p) * unit, .01 / unit)
    chain.init_obs_prec()
    chain.run_device(300)
    coef, _, ls, g = chain.get_state()
    tp = np.abs(coef[1:] / g) ** .5          # tilt^a with tilt = (beta/tau)^2
    qs = [0., .01, .1, .25, .5, .75, .9, .99, 1.]
    print("tilt^a quantiles", dict(zip(qs, np.quantile(tp, qs).round(3))))
    print("share in the plain-rejection regime (tilt^a < 2): %.3f" % (tp < 2).mean())
    print("expected plain-rejection trials, mean exp(tilt^a) over that regime: %.2f"
          % np.exp(tp[tp < 2]).mean())
