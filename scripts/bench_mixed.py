"""Times the products of a MIXED design at the config-3 shape: the synthetic
binary design of bench.py plus `n_cont` dense continuous columns (a fraction
of the binary entries can also carry values), tiled format.
Usage: python scripts/bench_mixed.py [n_cont] [reps] [valued_frac]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import scipy.sparse as sparse
import torch

from bayesbridge_amd import HipSparseDesignMatrix, simulate

n_cont = int(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
valued_frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.
n, p, f = 1000000, 50000, .002
indptr, indices = simulate.simulate_binary_csr_device(n, p, f, seed=111)
torch.cuda.synchronize()
Xb = sparse.csr_matrix((np.ones(indices.numel()), indices.cpu().numpy(),
                        indptr.cpu().numpy()), shape=(n, p))
rng = np.random.default_rng(3)
if valued_frac > 0:
    m = rng.random(Xb.nnz) < valued_frac
    Xb.data[m] = rng.standard_normal(int(m.sum()))
for label, X in (("all-binary", Xb if valued_frac == 0 else None),
                 ("mixed", sparse.hstack(
                     [Xb, sparse.csr_matrix(rng.standard_normal((n, n_cont)))]
                 ).tocsr() if n_cont else Xb)):
    if X is None:
        continue
    t0 = time.time()
    X.sort_indices()
    d = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                              storage='tiled')
    nn, P = d.shape
    v, w = rng.standard_normal(P), rng.standard_normal(nn)
    ref_v = v[0] + X @ v[1:] - np.asarray(X.mean(axis=0)).ravel() @ v[1:]
    err = np.abs(d.dot(v) - ref_v).max() / np.abs(ref_v).max()
    d.set_timing(True, every=1)
    d.reset_timing()
    for _ in range(reps):
        d.dot(v)
        d.Tdot(w)
    t = d.get_timing()
    db, tb = d.timed_bytes
    print("%-10s nnz=%d hybrid=%s built in %.1fs: dot %.4f ms (%.0f GB/s), "
          "tdot %.4f ms (%.0f GB/s), rel err %.1e"
          % (label, X.nnz, d.hybrid_info, time.time() - t0,
             t['dot'][1] / t['dot'][0], db / (t['dot'][1] / t['dot'][0]) / 1e6,
             t['tdot'][1] / t['tdot'][0],
             tb / (t['tdot'][1] / t['tdot'][0]) / 1e6, err))
    if os.environ.get("BENCH_MIXED_PAIR") == "1":
        # the K = 2 products of a batch on the same design (valued K-layout)
        from bayesbridge_amd import HipChainBatch, HipGibbsChain
        y = (rng.random(nn) < .3).astype(np.float64)
        chains = [HipGibbsChain(d, 'logit', y, n_trial=np.ones(nn), seed=s_)
                  for s_ in (1, 2)]
        batch = HipChainBatch(chains, allow_slow=True)
        V, W = rng.standard_normal((2, P)), rng.standard_normal((2, nn))
        T = batch.dot(V)
        batch.Tdot(W)
        errk = np.abs(T[0] - d.dot(V[0])).max() / np.abs(T[0]).max()
        d.reset_timing()
        for _ in range(20):
            batch.dot(V)
            batch.Tdot(W)
        t = d.get_timing()
        print("%-10s pair products: dot %.4f ms, tdot %.4f ms (both chains; "
              "the tiled kernels only: the dense block's kernels run outside the "
              "timers), rel err vs single %.1e"
              % (label, t['dot'][1] / t['dot'][0], t['tdot'][1] / t['tdot'][0], errk))
        # whole Gibbs iterations: one chain against the pair
        d.set_timing(False)
        for ch in chains:
            ch.set_state(global_scale=.01)
            ch.init_obs_prec()
        one = HipGibbsChain(d, 'logit', y, n_trial=np.ones(nn), seed=9)
        one.set_state(global_scale=.01)
        one.init_obs_prec()
        one.run(60, save=())
        t1 = time.time()
        r1 = one.run(40, save=())[0]
        t1 = time.time() - t1
        batch.run(60, save_coef=False)
        t2 = time.time()
        r2 = batch.run(40, save_coef=False)[0]
        t2 = time.time() - t2
        print("%-10s one chain %.1f it/s (n_cg %.1f); pair %.1f chain-it/s = %.2fx "
              "(n_cg %.1f)" % (label, 40 / t1, r1['n_cg_iter'].mean(), 80 / t2,
                               80 / t2 / (40 / t1), r2['n_cg_iter'].mean()))
        del batch, chains, one
    d.close() if hasattr(d, 'close') else None
    del d
