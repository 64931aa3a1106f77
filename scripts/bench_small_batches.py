"""Does a batch pay on SMALL designs?  Chain-iterations/s of HipChainBatch
against the same chains run one after the other, dense f32 and binary sparse.
Usage: python scripts/bench_small_batches.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
from bayesbridge_amd import (HipChainBatch, HipDenseDesignMatrix, HipGibbsChain,
                             HipSparseDesignMatrix, simulate)


def rate(run, iters):
    run(5)
    t0 = time.perf_counter()
    run(iters)
    return iters / (time.perf_counter() - t0)


rng = np.random.default_rng(0)
for n, p in ((2000, 500), (5000, 700), (20000, 2000), (50000, 4000)):
    X = rng.standard_normal((n, p)).astype(np.float32)
    y = X[:, :5].sum(axis=1) + rng.standard_normal(n)
    d = HipDenseDesignMatrix(X, center_predictor=True, add_intercept=True,
                             storage_dtype='float32')
    mk = lambda s: HipGibbsChain(d, 'linear', y, sd_unshrunk=[np.inf],
                                 bridge_exponent=.5, slab_size=2., seed=s)
    one = mk(1)
    r1 = rate(lambda k: one.run(k, save=()), 30)
    line = "dense %6d x %5d: one chain %7.1f it/s" % (n, p, r1)
    for K in (4, 16, 32):
        b = HipChainBatch([mk(10 + s) for s in range(K)], allow_slow=True)
        rk = K * rate(lambda k: b.run(k, save_coef=False), 15)
        line += "; K=%d %.2fx" % (K, rk / r1)
        del b
    print(line)
    del one, d
for n, p, f in ((5000, 500, .05), (20000, 2000, .02), (100000, 10000, .01)):
    X = simulate.simulate_binary_csr_fast(n, p, f, seed=3)
    y = (rng.random(n) < .3).astype(np.float64)
    d = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                              storage='tiled')
    def mk(s):
        ch = HipGibbsChain(d, 'logit', y, n_trial=np.ones(n), sd_unshrunk=[2.],
                           bridge_exponent=.5, slab_size=2., seed=s)
        ch.set_state(global_scale=.05)
        ch.init_obs_prec()
        return ch
    one = mk(1)
    r1 = rate(lambda k: one.run(k, save=()), 60)
    line = "sparse %6d x %5d: one chain %7.1f it/s" % (n, p, r1)
    for K in (2, 4):
        b = HipChainBatch([mk(10 + s) for s in range(K)], allow_slow=True)
        rk = K * rate(lambda k: b.run(k, save_coef=False), 30)
        line += "; K=%d %.2fx" % (K, rk / r1)
        del b
    print(line)
    del one, d
