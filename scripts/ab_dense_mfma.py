"""A/B of the dense GEMV on the vector ALUs (dense_dot_kernel) against the
matrix-core variant (dense_dot_mfma_kernel, BBX_DENSE_MFMA=1) at BASELINE
config 4 (200 000 x 8 000, f32 storage): checks that both give the same
product, then times them with HIP events.  Run once per variant:
    BBX_DENSE_MFMA=0|1 python scripts/ab_dense_mfma.py [n] [p]
(under `rocprofv3 --pmc ...` for the VALU / MFMA instruction counters)."""
import os
import sys
import time
from ctypes import c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR", os.path.join(ROOT, "bayes-bridge_amd")))
import torch
from bayesbridge_amd import HipDenseDesignMatrix, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
gen = torch.Generator(device="cuda")
gen.manual_seed(111)
X = torch.randn((n, p), generator=gen, device="cuda", dtype=torch.float32)
offset = X.double().mean(dim=0)
torch.cuda.synchronize()
design = HipDenseDesignMatrix.from_device_array(
    n, p, X.data_ptr(), offset.data_ptr(), add_intercept=True, device=0)
lib = _lib.load()
P = p + 1
v = torch.randn(P, dtype=torch.float64, device="cuda", generator=gen)
out = torch.empty(n, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
for _ in range(3):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()),
                           c_void_p(out.data_ptr()))
design.synchronize()
# reference on a row sample: torch f64 on the stored (f32-rounded) values
rows = torch.randint(0, n, (4000,), device="cuda", generator=gen)
A = (X[rows].double() - offset).float().double()
ref = v[0] + A @ v[1:]
err = float((out[rows] - ref).abs().max() / ref.abs().max())
design.set_timing(True)
design.reset_timing()
for _ in range(reps):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()),
                           c_void_p(out.data_ptr()))
cnt, ms = design.get_timing()["dot"]
bytes_ = design.matvec_bytes[0]
print("BBX_DENSE_MFMA=%s  dot %dx%d: %.4f ms  %.0f GB/s (%.1f%% of 8 TB/s)  "
      "max rel err vs torch f64 %.1e" % (
          os.environ.get("BBX_DENSE_MFMA", "0"), n, p, ms / cnt,
          bytes_ / (ms / cnt) / 1e6, bytes_ / (ms / cnt) / 1e6 / 80., err))
assert err < 1e-11
