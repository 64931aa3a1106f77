"""How much do independent chains / batches gain from running SIDE BY SIDE on
one GPU (one's ALU-bound sampler kernels under the other's bandwidth-bound CG
products)?  `procs` processes, each with its own design replica and a batch of
K chains (K = 1: a single chain), burn-in, then `steps` timed iterations
started together.  Prints aggregate chain-iterations/s.
Usage: python scripts/overlap_probe.py [procs] [K] [steps] [config]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, procs, K, steps, cfg, ready, go, out):
    sys.path.insert(0, os.path.join(ROOT, "bayes-bridge_amd"))
    os.environ["BBX_CHAIN_FORK"] = "0" if procs > 1 else os.environ.get(
        "BBX_CHAIN_FORK", "")
    if not os.environ["BBX_CHAIN_FORK"]:
        del os.environ["BBX_CHAIN_FORK"]
    import math
    import numpy as np
    import torch
    from bayesbridge_amd import (HipChainBatch, HipGibbsChain,
                                 HipSparseDesignMatrix)
    sys.path.insert(0, ROOT)
    import bench
    prob = bench.build_problem(torch, cfg, 111, "cuda:0")
    n, p, nnz = prob["n"], prob["p"], prob["nnz"]
    torch.cuda.synchronize()
    design = HipSparseDesignMatrix.from_device_csr(
        n, p, nnz, prob["indptr"].data_ptr(), prob["indices"].data_ptr(), None,
        prob["offset"].data_ptr(), add_intercept=True, device=0,
        storage="tiled")
    y = prob["n_success"].cpu().numpy()
    unit = math.gamma(2 / bench.ALPHA) / math.gamma(1 / bench.ALPHA)
    P = p + 1
    coef0 = np.zeros(P)
    ph = y.mean()
    coef0[0] = math.log(ph / (1 - ph))
    chains = []
    for i in range(K):
        ch = HipGibbsChain(design, 'logit', y, bridge_exponent=bench.ALPHA,
                           slab_size=bench.SLAB, seed=1000 * rank + i)
        ch.set_state(coef0, None, np.ones(P - 1) * unit, .01 / unit)
        ch.init_obs_prec()
        chains.append(ch)
    runner = HipChainBatch(chains, allow_slow=True) if K > 1 else chains[0]
    runner.run_device(150)                    # past most of the transient
    ready.put(rank)
    go.wait()
    t0 = time.perf_counter()
    res = runner.run_device(steps)
    torch.cuda.synchronize()
    out.put((rank, time.perf_counter() - t0, float(np.mean(res[2]))))


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cfg = sys.argv[4] if len(sys.argv) > 4 else "config3"
    ctx = mp.get_context("spawn")
    ready, out, go = ctx.Queue(), ctx.Queue(), ctx.Event()
    ps = [ctx.Process(target=worker, args=(r, procs, K, steps, cfg, ready, go,
                                           out)) for r in range(procs)]
    for p_ in ps:
        p_.start()
    for _ in ps:
        ready.get()
    go.set()
    res = [out.get() for _ in ps]
    for p_ in ps:
        p_.join()
    slowest = max(r[1] for r in res)
    print("procs=%d K=%d %s: %.1f chain-iterations/s in total (%.2f ms per "
          "process step, mean n_cg %.1f)"
          % (procs, K, cfg, procs * K * steps / slowest,
             1e3 * slowest / steps, sum(r[2] for r in res) / len(res)))
