"""The long-run distribution test (tests/test_hip_longrun.py) under OTHER seeds of
the device chain: max |z| and rms z of the means and variances per problem and
seed -- the test's seed (20261) is not a picked one.
    python scripts/longrun_seeds.py 1 2 3 > gpurun_out/r06_longrun_seeds.txt"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("bayes-bridge_amd", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import longrun_cases as lc
import test_hip_longrun as T

seeds = [int(a) for a in sys.argv[1:]] or [1, 2, 3]
golden = os.path.join(ROOT, "tests", "golden")
print("# device-RNG chain, %d kept iterations, against the reference fixtures; "
      "bounds of the test: |z| < %.1f, rms < 1.5" % (lc.DEV_KEEP, lc.Z_MAX))
for name in lc.CASES:
    case = lc.make_case(name)
    ref = T._fixture(golden, name, case)
    for seed in seeds:
        S, _, n_cg = T._device_series(case, seed=seed)
        zm, zv = lc.z_scores(lc.batch_stats([S]), ref)
        print("%-22s seed %-6d max|z| mean %.2f variance %.2f; rms %.2f / %.2f; "
              "mean n_cg %.2f" % (name, seed, np.abs(zm).max(), np.abs(zv).max(),
                                  np.sqrt((zm ** 2).mean()),
                                  np.sqrt((zv ** 2).mean()), n_cg.mean()),
              flush=True)
