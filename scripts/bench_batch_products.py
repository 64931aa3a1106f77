"""The batched (K-column) tiled products on their own at a bench.py config:
HIP-event timings of `reps` calls of HipChainBatch.dot / Tdot, against the
single-chain products of the same design.
Usage: python scripts/bench_batch_products.py [config3|config2] [K] [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import torch

# (before `bench`, which puts the repository's own package first on sys.path)
from bayesbridge_amd import HipChainBatch, HipGibbsChain, HipSparseDesignMatrix
import bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "config3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
prob = bench.build_problem(torch, cfg, 111, "cuda:0")
n, p, nnz = prob["n"], prob["p"], prob["nnz"]
torch.cuda.synchronize()
design = HipSparseDesignMatrix.from_device_csr(
    n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(), None,
    prob["offset"].data_ptr(), add_intercept=True, device=0, storage="tiled")
y = prob["n_success"].cpu().numpy()
chains = [HipGibbsChain(design, 'logit', y, seed=i) for i in range(K)]
for ch in chains:
    ch.init_obs_prec()
batch = HipChainBatch(chains, allow_slow=True)
rng = np.random.default_rng(1)
v, w = rng.standard_normal((K, p + 1)), rng.standard_normal((K, n))
batch.dot(v), batch.Tdot(w)
design.set_timing(True, every=1)
design.reset_timing()
for _ in range(reps):
    batch.dot(v)
    batch.Tdot(w)
t = design.get_timing()
db, tb = batch.launch_bytes
for name, nbytes in (("dot", db), ("tdot", tb)):
    cnt, ms = t[name]
    print("K=%d %-4s avg %.4f ms over %d launches, %.1f MB: %.0f GB/s "
          "(%.3f of 8 TB/s)" % (K, name, ms / cnt, cnt, nbytes / 1e6,
                                nbytes / (ms / cnt) / 1e6,
                                nbytes / (ms / cnt) / 1e6 / 8000))
