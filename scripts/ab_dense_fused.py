"""A/B of the single-pass dense operator g = X~^T (omega .* (X~ v)): the
register-prefetch kernel (dense_fused_kernel) against the LDS-DMA ring variants
(dense_fused_ring_kernel, BBX_DENSE_FUSED_RING=22 or unset from 64 rows per
workgroup on; 0 = register kernel everywhere).  The variants add the same numbers in the same order, so
the script prints a SHA-256 of the result next to the time: equal digests =
bit-identical products.  Run once per variant (the switch is read once per
process):
    BBX_DENSE_FUSED_RING=0|22 python scripts/ab_dense_fused.py [n] [p] [reps] [float32|float64]
(float64: the pair-layout kernels dense_fused_f64_kernel / _ring_kernel)
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import torch
from bayesbridge_amd import HipDenseDesignMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
storage = sys.argv[4] if len(sys.argv) > 4 else "float32"
gen = torch.Generator(device="cuda")
gen.manual_seed(111)
X = torch.randn((n, p), generator=gen, device="cuda", dtype=torch.float32)
offset = X.double().mean(dim=0)
torch.cuda.synchronize()
design = HipDenseDesignMatrix.from_device_array(
    n, p, X.data_ptr(), offset.data_ptr(), add_intercept=True, device=0,
    storage_dtype=storage)
del X
P = p + 1
rng = np.random.default_rng(4)
v = rng.standard_normal(P)
omega = rng.gamma(2., .5, n)
for _ in range(2):
    got = design.gram_matvec(omega, v)
two = design.Tdot(omega * design.dot(v))
err = float(np.abs(got - two).max() / np.abs(two).max())
design.set_timing(True)
design.reset_timing()
for _ in range(reps):
    again = design.gram_matvec(omega, v)
assert np.array_equal(again, got)
cnt, ms = design.get_timing()["dot"]       # the fused kernel is stamped as 'dot'
bytes_ = design.fused_operator_bytes
print("BBX_DENSE_FUSED_RING=%s  operator %dx%d %s: %.4f ms  %.0f GB/s (%.1f%% of "
      "8 TB/s)  launches %d  rel diff vs two products %.1e  sha256 %s" % (
          os.environ.get("BBX_DENSE_FUSED_RING", "default"), n, p, storage,
          ms / cnt,
          bytes_ / (ms / cnt) / 1e6, bytes_ / (ms / cnt) / 1e6 / 80., cnt, err,
          hashlib.sha256(got.tobytes()).hexdigest()[:16]))
assert err < 1e-10
