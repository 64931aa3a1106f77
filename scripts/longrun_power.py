"""How small a mis-scaling of the Polya-Gamma draws does the long-run test see?
The negative control of tests/test_hip_longrun.py (Omega multiplied by `scale`
between its draw and the next coefficient draw) at smaller scales, full length.
    python scripts/longrun_power.py 1.02 1.005 1.002 1.001"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("bayes-bridge_amd", "tests", os.path.join("tests", "golden")):
    sys.path.insert(0, os.path.join(ROOT, p))
import longrun_cases as lc
import test_hip_longrun as T

name = 'logit_mixed_ntrial'
case = lc.make_case(name)
ref = T._fixture(os.path.join(ROOT, "tests", "golden"), name, case)
print("# %s, %d kept iterations, Omega x scale; the test fails from |z| >= %.1f "
      "or rms >= 1.5" % (name, lc.DEV_KEEP, lc.Z_MAX))
for scale in [float(a) for a in sys.argv[1:]] or [1.02, 1.005, 1.002, 1.001]:
    S, _, _ = T._device_series(case, seed=20261, omega_scale=scale)
    zm, zv = lc.z_scores(lc.batch_stats([S]), ref)
    print("scale %.4f: max|z| mean %.2f variance %.2f; rms %.2f / %.2f; "
          "statistics beyond the bound: %d"
          % (scale, np.abs(zm).max(), np.abs(zv).max(),
             np.sqrt((zm ** 2).mean()), np.sqrt((zv ** 2).mean()),
             int((np.abs(zm) > lc.Z_MAX).sum() + (np.abs(zv) > lc.Z_MAX).sum())),
          flush=True)


def lambda_control(scale, keep=lc.DEV_KEEP):
    """The same harness with the LOCAL SCALES multiplied by `scale` between
    their draw and the next coefficient draw."""
    import warnings
    bridge = T._bridge(case)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bridge.gibbs(lc.BURNIN, n_burnin=lc.BURNIN, seed=20261,
                     init=dict(case['init']), coef_sampler_type='cg')
    chain = bridge._chain
    rows = {k: [] for k in ('coef', 'local_scale', 'obs_prec', 'global_scale',
                            'logp')}
    for _ in range(keep):
        _, _, ls, _ = chain.get_state()
        chain.set_state(local_scale=ls * scale)
        out, _ = chain.run(1, save=('coef', 'local_scale', 'obs_prec'))
        for k in rows:
            rows[k].append(out[k][0])
    s = {'coef': np.ascontiguousarray(np.array(rows['coef']).T),
         'local_scale': np.ascontiguousarray(np.array(rows['local_scale']).T),
         'global_scale': np.array(rows['global_scale']),
         'logp': np.array(rows['logp']),
         'obs_prec': np.ascontiguousarray(np.array(rows['obs_prec']).T)}
    bridge.prior.adjust_scale(s['global_scale'], s['local_scale'],
                              to='coef_magnitude')
    return lc.series(case, s)


if os.environ.get("BBX_POWER_LAMBDA"):
    print("# the same with the local scales multiplied by `scale` before the "
          "next coefficient draw")
    for scale in [float(a) for a in os.environ["BBX_POWER_LAMBDA"].split(",")]:
        S = lambda_control(scale)
        zm, zv = lc.z_scores(lc.batch_stats([S]), ref)
        print("lambda scale %.4f: max|z| mean %.2f variance %.2f; rms %.2f / "
              "%.2f; statistics beyond the bound: %d"
              % (scale, np.abs(zm).max(), np.abs(zv).max(),
                 np.sqrt((zm ** 2).mean()), np.sqrt((zv ** 2).mean()),
                 int((np.abs(zm) > lc.Z_MAX).sum()
                     + (np.abs(zv) > lc.Z_MAX).sum())), flush=True)
