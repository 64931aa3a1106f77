"""Random shapes / widths / storage types through the batched dense products,
against NumPy f64 on the stored values.  Usage: python scripts/fuzz_dense_batch.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
from bayesbridge_amd import HipChainBatch, HipDenseDesignMatrix, HipGibbsChain

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2024)
worst = 0.
for t in range(cases):
    n = int(rng.choice([5, 9, 15, 63, 64, 65, 127, 300, 1023, 4096, 5001, 17000]))
    n += int(rng.integers(0, 5))
    p = int(rng.choice([1, 2, 7, 15, 16, 31, 63, 64, 200, 255, 256, 257, 1000, 2049]))
    K = int(rng.choice([2, 4, 8, 16, 32]))
    storage = str(rng.choice(['float32', 'float64']))
    icpt = bool(rng.integers(0, 2))
    X = rng.standard_normal((n, p))
    if storage == 'float32':
        X = X.astype(np.float32).astype(np.float64)
    hip = HipDenseDesignMatrix(X, center_predictor=False, add_intercept=icpt,
                               storage_dtype=storage)
    P = p + int(icpt)
    y = rng.standard_normal(n)
    chains = [HipGibbsChain(hip, 'linear', y, sd_unshrunk=[np.inf] if icpt else [],
                            bridge_exponent=.5, slab_size=2., seed=s)
              for s in range(K)]
    batch = HipChainBatch(chains, allow_slow=True)
    V, W = rng.standard_normal((K, P)), rng.standard_normal((K, n))
    Xi = np.hstack([np.ones((n, 1)), X]) if icpt else X
    T, G = batch.dot(V), batch.Tdot(W)
    e1 = np.abs(T - V @ Xi.T).max() / max(np.abs(V @ Xi.T).max(), 1e-300)
    e2 = np.abs(G - W @ Xi).max() / max(np.abs(W @ Xi).max(), 1e-300)
    worst = max(worst, e1, e2)
    flag = "" if max(e1, e2) < 1e-11 else "   <-- FAIL"
    print("n=%6d p=%5d K=%2d %s icpt=%d: dot %.1e tdot %.1e%s"
          % (n, p, K, storage, icpt, e1, e2, flag))
    del batch, chains, hip
print("worst relative error %.2e" % worst)
sys.exit(0 if worst < 1e-11 else 1)
