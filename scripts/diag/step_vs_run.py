"""Diagnostic: is a device chain stepped one iteration at a time (with and
without a get_state / set_state round trip in between) bit for bit the chain
of one long run?"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import longrun_cases as lc
from bayesbridge_amd import BayesBridge, RegressionCoefPrior, RegressionModel

warnings.simplefilter("ignore")
name = sys.argv[1] if len(sys.argv) > 1 else 'logit_mixed_ntrial'
case = lc.make_case(name)
T = 40


def fresh():
    b = BayesBridge(RegressionModel(case['outcome'], case['X'].copy(),
                                    case['family']),
                    RegressionCoefPrior(**case['prior_kw']))
    b.gibbs(50, n_burnin=50, seed=5, init=dict(case['init']))
    return b, b._chain


save = ('coef', 'local_scale', 'obs_prec')
b, ch = fresh()
ref, _ = ch.run(T, save=save)
for mode in ('step', 'step+get', 'step+get+set_obs', 'step+get+set_all'):
    b, ch = fresh()
    rows = {k: [] for k in ref}
    for t in range(T):
        if mode != 'step':
            coef, obs, ls, g = ch.get_state()
            if mode == 'step+get+set_obs':
                ch.set_state(obs_prec=np.asarray(obs) * 1.0)
            elif mode == 'step+get+set_all':
                ch.set_state(coef, obs, ls, g)
        out, _ = ch.run(1, save=save)
        for k in rows:
            rows[k].append(out[k][0])
    msg = []
    for k in rows:
        a = np.array(rows[k]).reshape(ref[k].shape)
        bad = [t for t in range(T) if not np.array_equal(a[t], ref[k][t])]
        msg.append("%s: %s" % (k, "equal" if not bad else
                               "first differs at %d (max |d| %.2e)" % (
                                   bad[0], np.abs(a - ref[k]).max())))
    print(name, mode, "|", "; ".join(msg))
