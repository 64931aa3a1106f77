"""Times the batched dense products (dense_batch.hip) on their own: the design
timers bracket the kernels of `reps` calls of HipChainBatch.dot / Tdot.
Usage: python scripts/bench_dense_batch.py [n] [p] [K] [reps] [float32|float64]
Under `rocprofv3 --kernel-trace --stats` the per-kernel durations come out too."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("BBX_PACKAGE_DIR",
                                  os.path.join(ROOT, "bayes-bridge_amd")))
import numpy as np
import torch

from bayesbridge_amd import HipChainBatch, HipDenseDesignMatrix, HipGibbsChain

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
p = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
storage = sys.argv[5] if len(sys.argv) > 5 else 'float32'
gen = torch.Generator(device="cuda")
gen.manual_seed(111)
X = torch.randn((n, p), generator=gen, device="cuda", dtype=torch.float32)
off = X.double().mean(dim=0)
torch.cuda.synchronize()
design = HipDenseDesignMatrix.from_device_array(
    n, p, X.data_ptr(), off.data_ptr(), add_intercept=True, device=0,
    in_dtype='float32', storage_dtype=storage)
del X
y = np.random.default_rng(0).standard_normal(n)
chains = [HipGibbsChain(design, 'linear', y, seed=i) for i in range(K)]
batch = HipChainBatch(chains, allow_slow=True)
rng = np.random.default_rng(1)
v, w = rng.standard_normal((K, p + 1)), rng.standard_normal((K, n))
batch.dot(v), batch.Tdot(w)                       # warm-up
design.set_timing(True, every=1)
design.reset_timing()
for _ in range(reps):
    batch.dot(v)
    batch.Tdot(w)
t = design.get_timing()
db, tb = batch.launch_bytes
for name, nbytes in (("dot", db), ("tdot", tb)):
    cnt, ms = t[name]
    print("K=%d %-4s avg %.3f ms over %d launches: %.0f GB/s (%.3f of 8 TB/s), "
          "%.1f TFLOP/s f64 on the matrix cores (16 columns)"
          % (K, name, ms / cnt, cnt, nbytes / (ms / cnt) / 1e6,
             nbytes / (ms / cnt) / 1e6 / 8000,
             2. * n * (p + 8) * 16 / (ms / cnt) / 1e9))
# the same kernels inside the chains' CG loop (interleaved operands and results,
# row scale and <t, Omega t> in the epilogue): two Gibbs iterations of the batch
design.reset_timing()
design.set_timing(True, every=4)
samples, _ = batch.run(2, save_coef=False)
t = design.get_timing()
n_cg = samples['n_cg_iter'].max(axis=0).sum()
for name, nbytes in (("dot", db), ("tdot", tb)):
    cnt, ms = t[name]
    print("K=%d %-4s in the CG loop: avg %.3f ms over %d sampled launches "
          "(%.3f of 8 TB/s); %d lock-step CG iterations"
          % (K, name, ms / cnt, cnt, nbytes / (ms / cnt) / 1e6 / 8000, n_cg))
# single-chain reference on the same box: the one-pass operator
one = chains[0]
design.reset_timing()
design.set_timing(True, every=4)
s1 = one.run(2, save=())[0]
t = design.get_timing()
for name in t:
    cnt, ms = t[name]
    if cnt:
        print("one chain, %s launches: avg %.3f ms over %d sampled; n_cg %s"
              % (name, ms / cnt, cnt, s1['n_cg_iter']))
