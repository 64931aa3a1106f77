"""Effective GB/s of the OpenMP CPU baseline's products (oracle/omp_baseline.py)
by thread count on this host, on a 400k x 20k binary design.
Usage: python scripts/omp_baseline_probe.py [threads ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np

from bayesbridge_amd import simulate
from oracle.omp_baseline import OmpSparseDesign, load, usable_cores

print("cpu_count", os.cpu_count(), "affinity", usable_cores(),
      "omp_get_max_threads", load().oracle_omp_max_threads(),
      "OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"))
X = simulate.simulate_binary_csr_fast(400000, 20000, .005, seed=1)
v, w = np.ones(X.shape[1] + 1), np.ones(X.shape[0])
for T in [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128, 256]:
    d = OmpSparseDesign(X, n_threads=T)
    d.dot(v), d.Tdot(w)
    t0 = time.perf_counter()
    for _ in range(10):
        d.dot(v)
    t1 = time.perf_counter()
    for _ in range(10):
        d.Tdot(w)
    t2 = time.perf_counter()
    db, tb = d.product_bytes
    print("threads %3d: X v %.1f GB/s, X^T w %.1f GB/s (nnz %d)"
          % (T, db * 10 / (t1 - t0) / 1e9, tb * 10 / (t2 - t1) / 1e9, X.nnz))
    del d
