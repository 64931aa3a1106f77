"""Times the device scalar samplers per regime (run under rocprofv3
--kernel-trace and read dev_ts_kernel / dev_pg_kernel durations)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np
from bayesbridge_amd import _lib

lib = _lib.load()
n = 1 << 20
out = np.empty(n)
a = .25
for tp in (.1, 1., 1.9, 2.1, 4., 16., 100.):
    tilt = np.full(n, tp ** (1 / a))
    _lib.check(lib.bbx_device_tilted_stable(
        0, 1, n, a, tilt.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p)))
    print("ts tilt^a=%g mean=%g" % (tp, out.mean()))
shape = np.ones(n, dtype=np.int32)
for c in (0., 1., 5., 30.):
    tilt = np.full(n, c)
    _lib.check(lib.bbx_device_polya_gamma(
        0, 1, n, shape.ctypes.data_as(ctypes.c_void_p),
        tilt.ctypes.data_as(ctypes.c_void_p),
        out.ctypes.data_as(ctypes.c_void_p)))
    print("pg c=%g mean=%g" % (c, out.mean()))
