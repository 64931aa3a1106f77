"""Diagnostic: where do the two paths of
tests/test_hip_longrun.py::test_negative_control_passes_unscaled part?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("bayes-bridge_amd", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import longrun_cases as lc
import test_hip_longrun as T

for name in ('logit_mixed_ntrial', 'linear_dense'):
    case = lc.make_case(name)
    names = lc.series_names(case)
    S1, _, _ = T._device_series(case, seed=5, keep=300, omega_scale=1.0)
    S2, _, _ = T._device_series(case, seed=5, keep=300)
    S2 = S2[:300]
    bad = np.argwhere(S1 != S2)
    if len(bad) == 0:
        print(name, "equal")
        continue
    t0 = bad[:, 0].min()
    cols = sorted(set(bad[bad[:, 0] == t0][:, 1]))
    print(name, "first difference at kept iteration", t0, "in",
          [names[c] for c in cols][:8], "max |d| there",
          np.abs(S1[t0] - S2[t0]).max(), "; columns that ever differ:",
          sorted(set(names[c].split('[')[0] for c in set(bad[:, 1]))))
    print("  rel diff of global_scale at t0:",
          abs(S1[t0, names.index('log_global_scale')]
              - S2[t0, names.index('log_global_scale')]))
