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


def test_gpus_2_launches_two_ranks_by_itself():
    line = _bench("--gpus", "2", "--config", "tiny", "--steps", "4",
                  "--warmup", "1", "--burnin", "2")
    assert line["n_gpus"] == 2
    assert line["config"]["parallelism"] == "chains=2"
    assert line["config"]["backend"] in ("gloo", "nccl")
    assert line["cpu_baseline"] is None      # rank 0 at N = 1 only
