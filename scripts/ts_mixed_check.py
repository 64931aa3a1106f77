"""Device tilted-stable draws with MIXED tilts in one launch (like the chain's
local-scale update), binned by tilt^a: mean and quantiles of 1/sqrt(x)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
import numpy as np
from bayesbridge_amd import _lib
lib = _lib.load()
n = 2000000
rng = np.random.default_rng(0)
lo_, hi_ = (float(sys.argv[1]), float(sys.argv[2])) if len(sys.argv) > 2 else (.05, 9.)
tp = rng.uniform(lo_, hi_, n)
a = .25
tilt = tp ** (1 / a)
out = np.empty(n)
_lib.check(lib.bbx_device_tilted_stable(0, 5, n, a, tilt.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)))
lam = 1 / np.sqrt(out)
edges = list(np.linspace(lo_, hi_, 9))
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (tp >= lo) & (tp < hi)
    x = lam[m] * tilt[m] ** .5      # scale-free-ish
    print("tilt^a in [%.1f, %.1f): n=%d mean %.5f q01 %.5f q50 %.5f q99 %.5f" % (lo, hi, m.sum(), x.mean(), *np.quantile(x, [.01, .5, .99])))
