"""Random sparse designs (binary, binary + dense continuous columns, + a valued
rest, valued throughout) through the batched tiled products, against SciPy.
Usage: python scripts/fuzz_sparse_batch.py [cases]"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import scipy.sparse as sparse
from bayesbridge_amd import HipChainBatch, HipGibbsChain, HipSparseDesignMatrix

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(77)
worst = 0.
warnings.simplefilter("ignore")
for t in range(cases):
    n = int(rng.choice([40, 257, 1000, 5003, 20000, 70001]))
    p = int(rng.choice([3, 17, 64, 300, 2000, 9000]))
    dens = float(rng.choice([.002, .02, .2]))
    kind = str(rng.choice(['binary', 'dense_cols', 'rest', 'valued']))
    X = sparse.random(n, p, density=dens, format='csr', random_state=int(rng.integers(1 << 30)))
    X.data[:] = 1.
    # a few heavy rows / columns
    if n > 100 and p > 10:
        X = X.tolil()
        X[int(rng.integers(n)), :] = 1.
        X[rng.random(n) < .9, int(rng.integers(p))] = 1.
        X = X.tocsr()
    if kind in ('dense_cols', 'rest'):
        X = sparse.hstack([X, sparse.csr_matrix(rng.standard_normal((n, int(rng.integers(1, 4)))))]).tocsr()
    if kind == 'rest':
        m = (rng.random(X.nnz) < .1) & (X.data == 1.)
        X.data[m] = rng.standard_normal(int(m.sum()))
    if kind == 'valued':
        X.data[:] = rng.standard_normal(X.nnz)
    X.sort_indices()
    try:
        hip = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                                    storage='tiled')
    except Exception as e:       # e.g. every column constant
        print("case %d skipped: %s" % (t, str(e)[:80]))
        continue
    # the wrapper may have dropped constant columns: rebuild the reference from it
    n_, P = hip.shape
    keep = getattr(hip, 'kept_columns', None)
    Xc = X if keep is None else X[:, keep]
    if Xc.shape[1] + 1 != P:
        print("case %d skipped (columns dropped)" % t)
        continue
    off = np.asarray(Xc.mean(axis=0)).ravel()
    hy = hip.hybrid_info
    free = hip.is_binary or (hy is not None and hy['rest_nnz'] == 0)
    K = 4 if (free and rng.random() < .5) else 2
    y = (rng.random(n) < .4).astype(float)
    chains = [HipGibbsChain(hip, 'logit', y, n_trial=np.ones(n), sd_unshrunk=[2.],
                            slab_size=2., seed=s) for s in range(K)]
    batch = HipChainBatch(chains, allow_slow=True)
    V, W = rng.standard_normal((K, P)), rng.standard_normal((K, n))
    T, G = batch.dot(V), batch.Tdot(W)
    e = 0.
    for c in range(K):
        rt = V[c, 0] + Xc @ V[c, 1:] - off @ V[c, 1:]
        sw = W[c].sum()
        rg = np.concatenate([[sw], Xc.T @ W[c] - sw * off])
        e = max(e, np.abs(T[c] - rt).max() / max(np.abs(rt).max(), 1e-300),
                np.abs(G[c] - rg).max() / max(np.abs(rg).max(), 1e-300))
    worst = max(worst, e)
    print("n=%6d p=%5d dens=%.3f %-10s K=%d hybrid=%s: %.1e%s"
          % (n, p, dens, kind, K, hip.hybrid_info is not None, e,
             "" if e < 1e-10 else "   <-- FAIL"))
    del batch, chains, hip
print("worst relative error %.2e" % worst)
sys.exit(0 if worst < 1e-10 else 1)
