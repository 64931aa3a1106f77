"""Times the operator kernels alone (HIP events) on synthetic binary designs.
Usage: python scripts/bench_spmv.py [config2|config3|NxPxF] [csr|tiled] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (scripts/ab_spmv.sh points this at a private copy holding a build variant)
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))

import numpy as np
import torch

from bayesbridge_amd import HipSparseDesignMatrix, simulate, _lib
from ctypes import c_void_p

cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
storage = sys.argv[2] if len(sys.argv) > 2 else "csr"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
shapes = {"config2": (100000, 10000, .01),
           "config3": (1000000, 50000, .002),
           # same tiles as config3, twice as many per workgroup (marginal
           # cost of a tile vs per-launch fixed cost)
           "wide": (1000000, 100000, .002),
           "tall": (2000000, 50000, .002)}
if cfg in shapes:
    n, p, f = shapes[cfg]
else:                       # e.g. 400000x20000x0.005
    n, p, f = cfg.split("x")
    n, p, f = int(n), int(p), float(f)
t0 = time.time()
indptr, indices = simulate.simulate_binary_csr_device(n, p, f, seed=111)
torch.cuda.synchronize()
nnz = indices.numel()
print("generated %s: n=%d p=%d nnz=%d in %.1fs" % (cfg, n, p, nnz,
                                                   time.time() - t0))
offset = torch.bincount(indices.long(), minlength=p).double() / n
t0 = time.time()
design = HipSparseDesignMatrix.from_device_csr(
    n, p, nnz, indptr.data_ptr(), indices.data_ptr(), None, offset.data_ptr(),
    add_intercept=True, device=0, storage=storage)
print("design built in %.1fs, storage %.1f MB, format %s" % (
    time.time() - t0, design.storage_bytes / 1e6, design.storage_format))
if design.storage_format == 'tiled':
    print("tiled geometry:", design.tiled_info())
lib = _lib.load()
P = p + 1
v = torch.randn(P, dtype=torch.float64, device='cuda')
w = torch.randn(n, dtype=torch.float64, device='cuda')
out_n = torch.empty(n, dtype=torch.float64, device='cuda')
out_P = torch.empty(P, dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
for _ in range(5):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()),
                           c_void_p(out_n.data_ptr()))
    lib.bbx_design_tdot_dev(design.handle, c_void_p(w.data_ptr()),
                            c_void_p(out_P.data_ptr()))
design.synchronize()
design.set_timing(True)
design.reset_timing()
for _ in range(reps):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()),
                           c_void_p(out_n.data_ptr()))
    lib.bbx_design_tdot_dev(design.handle, c_void_p(w.data_ptr()),
                            c_void_p(out_P.data_ptr()))
tm = design.get_timing()
db, tb = design.matvec_bytes
for name, b in (("dot", db), ("tdot", tb)):
    cnt, ms = tm[name]
    avg = ms / cnt
    print("%-5s avg %.4f ms  algorithmic %.1f MB  -> %.1f GB/s (%.1f%% of 8 TB/s)"
          % (name, avg, b / 1e6, b / avg / 1e6, b / avg / 1e6 / 80.))
# check against a torch reference
ref = torch.zeros(n, dtype=torch.float64, device='cuda')
rows = torch.repeat_interleave(torch.arange(n, device='cuda'),
                               (indptr[1:] - indptr[:-1]).long())
ref.index_add_(0, rows, v[1:][indices.long()])
ref += v[0] - torch.dot(offset, v[1:])
print("dot  max abs err vs torch:", float((ref - out_n).abs().max()))
refT = torch.zeros(p, dtype=torch.float64, device='cuda')
refT.index_add_(0, indices.long(), w[rows])
refT -= w.sum() * offset
print("tdot max abs err vs torch:", float((refT - out_P[1:]).abs().max()),
      float((out_P[0] - w.sum()).abs()))
