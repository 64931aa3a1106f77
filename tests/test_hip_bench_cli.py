"""GPU: bench.py's contract end to end on a small configuration -- one JSON
line with the roofline / cpu_baseline objects, and `--gpus 2` launching two
ranks by itself (on a 1-GPU box they share the device and gather over gloo;
on an 8-GPU node the same path runs one rank per GPU over RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")]
                         + list(flags), env=env, capture_output=True,
                         text=True, timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_single_gpu_line_has_roofline_and_both_cpu_baselines():
    line = _bench("--config", "tiny", "--steps", "6", "--warmup", "2",
                  "--burnin", "4", "--cpu-baseline-iters", "5")
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["warmup"] == 2
    assert line["unit"] == "Gibbs iters/sec" and line["value"] > 0
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-2 * line["value"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["operator_frac"] < 1 and 0 < r["iteration_frac"] < 1
    assert r["operator"]["launches"] > 0
    assert r["other"]["tdot"]["bytes"] > 0
    for key in ("cpu_baseline", "cpu_baseline_omp"):
        b = line[key]
        assert b["value"] > 0 and b["cores"] >= 1
        assert b["sample"].startswith("5 Gibbs iterations")
    assert line["cpu_baseline"]["kind"] == "port"
    assert line["cpu_baseline_omp"]["kind"] == "port-omp"
    assert line["cpu_baseline_omp"]["dot_gbs"] > 0
    # k chains per GPU sharing the passes over X; `value` stays the one-chain rate
    mc = line["multi_chain"]
    assert abs(mc["k=1"]["chain_iters_per_sec"] - line["value"]) < 1e-2 * line["value"]
    for k in ("k=2", "k=4"):
        assert mc[k]["chain_iters_per_sec"] > 0 and mc[k]["vs_k1"] > 0
        assert mc[k]["dot"]["bytes"] > 0 and mc[k]["tdot"]["launches"] > 0
    assert line["config"]["startup_s"] > 0 and line["config"]["peak_host_rss_mb"] > 0


def test_roofline_traffic_is_measured_in_the_run():
    """`roofline.traffic` of the sparse BASELINE configs comes from two
    rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that bench.py runs
    itself after the timed region, over the product kernels on the same matrix
    and layout -- not from a committed profile.  Config 2: the launch's HBM
    bytes within 10 % of its algorithmic bytes (part of the 26 MB is served by
    the Infinity Cache)."""
    line = _bench("--config", "config2", "--steps", "20", "--warmup", "5",
                  "--burnin", "30", "--cpu-baseline-iters", "0",
                  "--multi-chain", "0", "--repeat", "1")
    r = line["roofline"]
    assert r["traffic_source"].startswith("measured in this run"), \
        r["traffic_source"]
    assert .85 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 1.15
    off = _bench("--config", "config2", "--steps", "5", "--warmup", "2",
                 "--burnin", "5", "--cpu-baseline-iters", "0",
                 "--multi-chain", "0", "--repeat", "1", "--live-traffic", "0")
    assert off["roofline"]["traffic"] is None      # no committed config-2 profile


def test_dense_line_has_a_cpu_baseline():
    """config 4's shape in miniature: the dense line carries `cpu_baseline`
    (NumPy dgemv products + SciPy cg, BLAS threads stated), like the sparse
    line carries the SciPy CSR one."""
    line = _bench("--config", "tiny-dense", "--steps", "6", "--warmup", "2",
                  "--burnin", "4", "--multi-chain", "4")
    assert line["value"] > 0 and line["roofline"]["bound"] == "hbm"
    assert "dense" in line["config"]["workload"]
    b = line["cpu_baseline"]
    assert b["kind"] == "port" and b["value"] > 0 and b["cores"] >= 1
    assert b["sample"].startswith("2 Gibbs iterations")
    assert "dgemv" in b["sample"] and b["blas"]
    assert b["dot_gbs"] > 0 and b["tdot_gbs"] > 0
    assert line["multi_chain"]["k=4"]["chain_iters_per_sec"] > 0


def test_gpus_2_launches_two_ranks_by_itself():
    line = _bench("--gpus", "2", "--config", "tiny", "--steps", "4",
                  "--warmup", "1", "--burnin", "2")
    assert line["n_gpus"] == 2
    assert line["config"]["parallelism"] == "chains=2"
    assert line["config"]["backend"] in ("gloo", "nccl")
    assert line["cpu_baseline"] is None      # rank 0 at N = 1 only


def test_launcher_with_one_rank_runs_the_rccl_path():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: under a
    launcher bench.py creates a process group even for one rank, so on a 1-GPU
    box RCCL is initialised and the N-rank path's collectives -- the gather of
    the samples, the MAX all-reduce of the time, the barriers, the
    one-device-per-rank check -- run over `backend='nccl'` (config 5's path
    with world size 1)."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
         "--master-port", "29641", os.path.join(ROOT, "bench.py"),
         "--gpus", "1", "--config", "tiny", "--steps", "4", "--warmup", "1",
         "--burnin", "2", "--cpu-baseline-iters", "0"],
        env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1
    assert line["config"]["backend"] == "nccl"
    assert line["config"]["rccl_ranks"] == 1


def test_rccl_collectives_of_the_chain_gather_with_one_rank():
    """chains.gather_chain_samples / max_over_ranks / barrier on a world-size-1
    `nccl` group: device tensors stay on the device, host tensors are moved
    there (RCCL moves device memory only) and come back on the host."""
    code = """
import os, sys
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from bayesbridge_amd import chains
rank, world, local = chains.init_process_group_from_env(
    backend='nccl', single_rank_group=True)
assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl'
x = torch.arange(12, dtype=torch.float64, device='cuda').reshape(3, 4)
got = chains.gather_chain_samples(x, dst=0)
assert got.is_cuda and got.shape == (1, 3, 4) and torch.equal(got[0], x)
host = torch.arange(6, dtype=torch.float64).reshape(2, 3)
got = chains.gather_chain_samples(host, dst=0)
assert not got.is_cuda and torch.equal(got[0], host)
assert chains.max_over_ranks(2.5) == 2.5
chains.barrier()
ids = [None]
dist.all_gather_object(ids, torch.cuda.current_device())
assert ids == [0]
dist.destroy_process_group()
print('RCCL_OK')
""" % os.path.join(ROOT, "bayes-bridge_amd")
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29643",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RCCL_OK" in out.stdout


def test_gpus_8_dry_run_over_gloo():
    """The driver's 8-GPU launch in miniature: `bench.py --gpus 8` starts eight
    ranks by itself (ports, eight generators and eight host-side layout
    builders side by side); on a 1-GPU box they share the device (one stream
    per chain, BBX_CHAIN_FORK=0) and gather over gloo."""
    line = _bench("--gpus", "8", "--config", "tiny", "--steps", "3",
                  "--warmup", "1", "--burnin", "2", timeout=1500)
    assert line["n_gpus"] == 8
    assert line["config"]["parallelism"] == "chains=8"
    assert line["value"] > 0
    assert line["config"]["startup_s"] > 0
    assert line["config"]["peak_host_rss_mb"] > 0
    # the line explains itself rank by rank (timed seconds before the MAX
    # all-reduce, the chain alone, the gather, n_cg, the device each sat on)
    pr = line["config"]["per_rank"]
    for key in ("startup_s", "peak_host_rss_mb", "builder_threads", "timed_s",
                "run_s", "gather_ms", "iters_per_sec", "mean_n_cg_iter",
                "burnin_ms_per_step", "device_index", "device_name", "pid"):
        assert len(pr[key]) == 8, key
    assert all(v > 0 for v in pr["timed_s"]) and all(v > 0 for v in pr["run_s"])
    assert all(r <= t + 1e-3 for r, t in zip(pr["run_s"], pr["timed_s"]))
    assert all(v >= 0 for v in pr["gather_ms"])
    assert len(set(pr["pid"])) == 8
    # `value` is N * K / the slowest rank's timed seconds
    assert abs(line["value"] - 8 * 3 / max(pr["timed_s"])) < 2e-2 * line["value"]
