"""Soak test of the product kernels: the same operator application
X~^T (Omega (X~ v)) launched N times back to back must give the SAME bits every
time -- the tiled kernels keep a hand-counted ring of asm-issued loads
(s_waitcnt vmcnt(k)), the dense-block kernels exchange sums through LDS: a
hazard that shows once in 10^4 launches is silent corruption of a chain.
Usage: python scripts/soak_products.py config3|config2|mixed|mixed_small|dense [N] [perturb]
(`perturb`: a second stream runs bursts of unrelated kernels -- a GEMM, a sort
of varying length -- beside the products, so that waves are scheduled, delayed
and resumed differently from launch to launch)
Prints the number of launches whose output differed from the first (0 = clean)."""
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

from bayesbridge_amd import (HipDenseDesignMatrix, HipSparseDesignMatrix, _lib,
                             simulate)

what = sys.argv[1] if len(sys.argv) > 1 else "config3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
t0 = time.time()
if what == "config3":
    ip, ix = simulate.simulate_binary_csr_device(1000000, 50000, .002, seed=111)
    d = HipSparseDesignMatrix.from_device_csr(
        1000000, 50000, int(ix.numel()), ip.data_ptr(), ix.data_ptr(),
        add_intercept=True)
elif what == "config2":
    X = simulate.simulate_binary_csr_fast(100000, 10000, .01, seed=111)
    d = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True)
elif what in ("mixed", "mixed_small"):
    rng = np.random.default_rng(3)
    rows, nb, nd = (200000, 3000, 1500) if what == "mixed" else (60000, 1000, 300)
    Xb = simulate.simulate_binary_csr_fast(rows, nb, .02, seed=9)
    X = sparse.hstack([Xb, sparse.csr_matrix(rng.standard_normal((rows, nd)))]).tocsr()
    X.sort_indices()
    d = HipSparseDesignMatrix(X, center_predictor=True, add_intercept=True,
                              storage='tiled')
else:
    rng = np.random.default_rng(3)
    d = HipDenseDesignMatrix(rng.standard_normal((100000, 4000)),
                             center_predictor=True, add_intercept=True,
                             storage_dtype='float32')
n, P = d.shape
lib = _lib.load()
gen = torch.Generator(device='cuda')
gen.manual_seed(1)
v = torch.randn(P + 1, dtype=torch.float64, device='cuda', generator=gen)[1:]
om = torch.rand(n, dtype=torch.float64, device='cuda', generator=gen) + .1
first = torch.empty(P, dtype=torch.float64, device='cuda')
out = torch.empty(P, dtype=torch.float64, device='cuda')
bad = torch.zeros(1, dtype=torch.int64, device='cuda')


def apply(dst):
    _lib.check(lib.bbx_design_gram_matvec_dev(
        d.handle, c_void_p(om.data_ptr()), c_void_p(v.data_ptr()),
        c_void_p(dst.data_ptr())))


apply(first)
d.synchronize()
assert bool(torch.all(torch.isfinite(first)))
print("%s: design %dx%d (%s) ready in %.0f s; %d launches ..."
      % (what, n, P, d.storage_format, time.time() - t0, N), flush=True)
perturb = len(sys.argv) > 3
side = torch.cuda.Stream()
ga = torch.randn(2048, 2048, device='cuda')
junk = torch.randn(1 << 20, device='cuda')
t0 = time.time()
for k in range(N):
    if perturb and k % 3 == 0:
        with torch.cuda.stream(side):
            if k % 2:
                ga = (ga @ ga).clamp_(-1., 1.)
            else:
                junk[: 4096 + 97 * (k % 1000)].sort()
    apply(out)
    d.synchronize()               # the compare runs on torch's stream
    bad += (out != first).any().to(torch.int64)
torch.cuda.synchronize()
print("%s: %d of %d operator applications differ from the first (%.1f s)"
      % (what, int(bad.item()), N, time.time() - t0))
sys.exit(1 if int(bad.item()) else 0)
