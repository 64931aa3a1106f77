"""Runs a batch of K chains at a bench.py config for a few Gibbs iterations
(meant to run under `rocprofv3 --kernel-trace`), or, with --analyse DIR,
condenses the trace: per batch step, the time inside the CG loop (first to last
product kernel), between two CG loops (pre/post-draw kernels), and idle.
Usage: python scripts/batch_timeline.py config3 2 40
       python scripts/batch_timeline.py --analyse gpurun_out/bt_trace"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))

if sys.argv[1] == "--analyse":
    import numpy as np
    import pandas as pd
    f = sorted(glob.glob(sys.argv[2] + "/*/*kernel_trace.csv"))[-1]
    d = pd.read_csv(f).sort_values("Start_Timestamp")
    name = d.Kernel_Name.str.replace("void ", "").str.split("(").str[0]
    d = d.assign(short=name, dur=d.End_Timestamp - d.Start_Timestamp)
    # the timed part: after the last b_setup-less gap... use the b_setup kernels
    setups = d[d.short.str.contains("b_setup_kernel")]
    t_setup = setups.Start_Timestamp.values
    steps = []
    for a, b in zip(t_setup[5:-1], t_setup[6:]):      # skip warm-up steps
        s = d[(d.Start_Timestamp >= a) & (d.Start_Timestamp < b)]
        prod = s[s.short.str.contains("tiled_spmv_kernel|dense_dot_kd|dense_tdot_kd")]
        cg_end = prod.End_Timestamp.values[-2]        # last product = linear predictor
        busy = s.dur.sum()
        steps.append(dict(step=(b - a) / 1e3, cg=(cg_end - a) / 1e3,
                          rest=(b - cg_end) / 1e3, busy=busy / 1e3,
                          n_prod=len(prod)))
    df = pd.DataFrame(steps)
    print("batch steps analysed: %d" % len(df))
    print(df.mean().round(1).to_string())
    # kernels of the part outside the CG loop, summed per step
    a, b = t_setup[-2], t_setup[-1]
    s = d[(d.Start_Timestamp >= a) & (d.Start_Timestamp < b)]
    prod = s[s.short.str.contains("tiled_spmv_kernel")]
    tail = s[s.Start_Timestamp >= prod.End_Timestamp.values[-2]]
    print("outside the CG loop (one step), kernel: start offset us, duration us, stream")
    t0 = tail.Start_Timestamp.values[0]
    for _, r in tail.iterrows():
        print("  %-34s %8.1f %8.1f  q%s" % (r.short[:34], (r.Start_Timestamp - t0) / 1e3,
                                            r.dur / 1e3, r.get("Queue_Id", "")))
    sys.exit(0)

import numpy as np
import torch
from bayesbridge_amd import HipChainBatch, HipGibbsChain, HipSparseDesignMatrix
import bench

cfg, K, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prob = bench.build_problem(torch, cfg, 111, "cuda:0")
n, p, nnz = prob["n"], prob["p"], prob["nnz"]
torch.cuda.synchronize()
design = HipSparseDesignMatrix.from_device_csr(
    n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(), None,
    prob["offset"].data_ptr(), add_intercept=True, device=0, storage="tiled")
y = prob["n_success"].cpu().numpy()
chains = [HipGibbsChain(design, 'logit', y, n_trial=np.ones(n), sd_unshrunk=[np.inf],
                        bridge_exponent=.5, slab_size=2., seed=111 + i)
          for i in range(K)]
for ch in chains:
    ch.set_state(global_scale=.01)
    ch.init_obs_prec()
batch = HipChainBatch(chains, allow_slow=True)
s, _ = batch.run(iters, save_coef=False)
print("n_cg", s["n_cg_iter"][:, -5:])
