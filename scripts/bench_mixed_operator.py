"""One operator application X~^T (Omega (X~ v)) on MIXED designs of the
reference's test shape (tests/helper.py:13, simulate_data.py:29-63:
simulate_design(n, p, binary_frac=.9) -- 10 % dense Gaussian columns, the rest
binary with frequency `freq`): torch events around bbx_design_gram_matvec_dev,
the single pass over the dense block (hyb_dense_fused_kernel) against the two
separate kernels (BBX_HYB_FUSED=0).
Usage: python scripts/bench_mixed_operator.py n p binary_frac freq [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import scipy.sparse as sparse
import torch
from ctypes import c_void_p

from bayesbridge_amd import HipSparseDesignMatrix, _lib, simulate

n, p = int(sys.argv[1]), int(sys.argv[2])
binary_frac, freq = float(sys.argv[3]), float(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 50
n_bin = int(round(p * binary_frac))
n_cont = p - n_bin
rng = np.random.default_rng(3)
t0 = time.time()
Xb = simulate.simulate_binary_csr_fast(n, n_bin, freq, seed=9)
X = sparse.hstack([Xb, sparse.csr_matrix(rng.standard_normal((n, n_cont)))]).tocsr()
X.sort_indices()
d = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                          storage='tiled')
info = d.hybrid_info
nn, P = d.shape
print("design %dx%d: %d binary (nnz %d) + %d dense columns, built in %.0f s, "
      "hybrid %s" % (n, p, n_bin, Xb.nnz, n_cont, time.time() - t0, info))
lib = _lib.load()
# (v + 1 double, the value-free kernel's input behind the intercept, must be
# 16-byte aligned for the fused forms, as the CG loop's own buffers are)
v = torch.randn(P + 1, dtype=torch.float64, device='cuda')[1:]
om = torch.rand(nn, dtype=torch.float64, device='cuda') + .1
out = torch.empty(P, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()


def apply():
    _lib.check(lib.bbx_design_gram_matvec_dev(
        d.handle, c_void_p(om.data_ptr()), c_void_p(v.data_ptr()),
        c_void_p(out.data_ptr())))


for _ in range(5):
    apply()
d.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    apply()
d.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / reps
# check against NumPy
off = np.asarray(X.mean(axis=0)).ravel()
vh, oh = v.cpu().numpy(), om.cpu().numpy()
t = vh[0] + X @ vh[1:] - off @ vh[1:]
wv = oh * t
ref = np.concatenate(([wv.sum()], X.T @ wv - wv.sum() * off))
err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
dense_b = 8. * nn * n_cont
bin_b = 2 * 2.2 * Xb.nnz      # both orientations of the value-free part, ~2.2 B/entry
floor_ms = (dense_b + bin_b) / 6e12 * 1e3
print("operator application %.4f ms (rel err %.1e); D once + binary part at "
      "6 TB/s = %.4f ms -> %.2f x that; D = %.0f MB"
      % (ms, err, floor_ms, ms / floor_ms, dense_b / 1e6))
