"""BASELINE config 4: linear model, dense X 200k x 8k stored in f32 on one
MI355X (generated on the device with torch's Philox generator, seed 111),
centred + intercept.  Times the dense dot/Tdot kernels (HIP events) and a few
Gibbs iterations of the device chain."""
import ctypes
import json
import os
import sys
import time
from ctypes import byref, c_double, c_void_p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np
import torch
from bayesbridge_amd import _lib
from bayesbridge_amd.design_matrix import HipDenseDesignMatrix, HipDesignMatrix

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = "cuda:0"
gen = torch.Generator(device=dev)
gen.manual_seed(111)
X = torch.randn((n, p), generator=gen, device=dev, dtype=torch.float32)
offset = X.double().mean(dim=0)
beta = torch.zeros(p, dtype=torch.float64, device=dev)
beta[:5], beta[5:10], beta[10:15] = 1.5, 1., .5
y = (X[:, :15].double() @ beta[:15]) + torch.randn(n, generator=gen, device=dev,
                                                   dtype=torch.float64)
lib = _lib.load()
design = HipDenseDesignMatrix.__new__(HipDenseDesignMatrix)
HipDesignMatrix.__init__(design)
design.centered, design.intercept_added, design.column_offset = True, True, None
t0 = time.time()
_lib.check(lib.bbx_design_create_dense_dev(
    n, p, c_void_p(X.data_ptr()), _lib.F32, _lib.F32,
    c_void_p(offset.data_ptr()), 1, 0, byref(design._h)))
torch.cuda.synchronize()
print("dense design built in %.2fs" % (time.time() - t0))
del X
P = p + 1
v = torch.randn(P, dtype=torch.float64, device=dev)
w = torch.randn(n, dtype=torch.float64, device=dev)
on = torch.empty(n, dtype=torch.float64, device=dev)
oP = torch.empty(P, dtype=torch.float64, device=dev)
for _ in range(3):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()), c_void_p(on.data_ptr()))
    lib.bbx_design_tdot_dev(design.handle, c_void_p(w.data_ptr()), c_void_p(oP.data_ptr()))
design.synchronize()
design.set_timing(True)
design.reset_timing()
for _ in range(20):
    lib.bbx_design_dot_dev(design.handle, c_void_p(v.data_ptr()), c_void_p(on.data_ptr()))
    lib.bbx_design_tdot_dev(design.handle, c_void_p(w.data_ptr()), c_void_p(oP.data_ptr()))
tm = design.get_timing()
design.set_timing(False)
db, tb = design.matvec_bytes
res = {}
for name, b in (("dot", db), ("tdot", tb)):
    cnt, ms = tm[name]
    res[name] = dict(avg_ms=ms / cnt, bytes=b, gbs=b / (ms / cnt) / 1e6)
    print("%-5s avg %.4f ms  %.1f MB -> %.1f GB/s (%.1f%% of 8 TB/s)" % (
        name, ms / cnt, b / 1e6, res[name]["gbs"], res[name]["gbs"] / 80.))
# chain: linear model, demo prior/init
chain = c_void_p()
yh = y.cpu().numpy()
sd = np.array([np.inf])
_lib.check(lib.bbx_chain_create(design.handle, _lib.MODEL_LINEAR,
                                yh.ctypes.data_as(c_void_p), None, 1,
                                sd.ctypes.data_as(c_void_p), .5, 2., 0., 0., 111,
                                byref(chain)))
import math
unit = math.gamma(4.) / math.gamma(2.)
coef0 = np.zeros(P)
coef0[0] = yh.mean()
ls0 = np.ones(p) * unit
g0 = c_double(.01 / unit)
_lib.check(lib.bbx_chain_set_state(chain, coef0.ctypes.data_as(c_void_p), None,
                                   ls0.ctypes.data_as(c_void_p), byref(g0)))
_lib.check(lib.bbx_chain_init_obs_prec(chain))
ncg = np.zeros(steps)
_lib.check(lib.bbx_chain_run(chain, 3, 0, 1, 500, 0., None, None, None, None, None, None))
torch.cuda.synchronize()
t0 = time.perf_counter()
_lib.check(lib.bbx_chain_run(chain, steps, 0, 1, 500, 0., None, None, None, None, None,
                             ncg.ctypes.data_as(c_void_p)))
dt = time.perf_counter() - t0
print(json.dumps(dict(config="config4 dense %dx%d f32" % (n, p),
                      gibbs_it_per_s=steps / dt, ms_per_iter=1e3 * dt / steps,
                      mean_n_cg=float(ncg.mean()), kernels=res)))
lib.bbx_chain_destroy(chain)
