"""Soak test of the device chain: the same chain (same seed, same start) run
twice for N iterations must give the same samples bit for bit -- two streams
after every draw, work enqueued speculatively behind the CG loop's stop test,
normals filled ahead: a race between them would show as a difference.
Usage: python scripts/soak_chain.py config3|config2 [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import torch

from bayesbridge_amd import HipGibbsChain, HipSparseDesignMatrix, simulate

what = sys.argv[1] if len(sys.argv) > 1 else "config2"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
n, p, f = (1000000, 50000, .002) if what == "config3" else (100000, 10000, .01)
ip, ix = simulate.simulate_binary_csr_device(n, p, f, seed=111)
d = HipSparseDesignMatrix.from_device_csr(n, p, int(ix.numel()), ip.data_ptr(),
                                          ix.data_ptr(), add_intercept=True)
y = (np.random.default_rng(5).random(n) < .2).astype(np.float64)
runs = []
for rep in range(2):
    chain = HipGibbsChain(d, 'logit', y, sd_unshrunk=[2.], slab_size=1., seed=11)
    coef0 = np.zeros(p + 1)
    coef0[0] = np.log(.2 / .8)
    chain.set_state(coef0, None, np.ones(p), .01)
    chain.init_obs_prec()
    t0 = time.time()
    out, bad = chain.run(N, save=('coef', 'local_scale'))
    runs.append(out)
    print("%s run %d: %d iterations in %.1f s, mean n_cg %.1f, unconverged %d"
          % (what, rep, N, time.time() - t0, out['n_cg_iter'].mean(), bad),
          flush=True)
    chain.close()
    # churn the allocators between the runs
    junk = [torch.randn(1 << 22, device='cuda') for _ in range(4)]
    del junk
same = all(np.array_equal(runs[0][k], runs[1][k]) for k in runs[0])
print("%s: two runs of %d iterations %s" % (
    what, N, "agree bit for bit (coef, local_scale, global_scale, logp, n_cg)"
    if same else "DIFFER"))
if os.environ.get("BBX_SOAK_SAVE"):
    # (scripts/r06_soak.sh compares the samples ACROSS host-side variants)
    np.savez(os.environ["BBX_SOAK_SAVE"], **{
        k: (v[::50] if v.ndim > 1 else v) for k, v in runs[0].items()})
sys.exit(0 if same else 1)
