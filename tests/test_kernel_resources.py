"""CPU (hipcc cross-compile, ~30 s): the LDS-tiled kernel issues its stream
loads through inline asm with hand-counted waits, so the compiler must not
spill or use scratch there (a spilled register that is the destination of an
in-flight asm load would be silent corruption), and the 1024-thread workgroup
must fit the 128-VGPR budget."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _resource_table(src, tmp_path):
    out = subprocess.run(
        [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950",
         "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o",
         str(tmp_path / "t.o")],
        capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    # "remark: Function Name: <mangled>" then one "remark:     <key>: <value>"
    # line per resource
    current, table = None, {}
    for line in out.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            current = m.group(1)
            table[current] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][A-Za-z ]*?(?: \[[^\]]*\])?): (\d+)", line)
        if m and current:
            table[current][m.group(1)] = int(m.group(2))
    return table


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_tiled_kernels_do_not_spill(tmp_path):
    table = _resource_table(
        os.path.join(ROOT, "bayes-bridge_amd", "csrc", "spmv_tiled.hip"),
        tmp_path)
    tiled = {k: v for k, v in table.items() if "tiled_spmv_kernel" in k}
    # value-free and valued, each with 8- and 16-byte
    # slice refills
    assert len(tiled) >= 4
    for name, res in tiled.items():
        assert res["VGPRs"] <= 128, (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["SGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_dense_fused_kernels_fit_the_register_budget(tmp_path):
    """The single-pass dense operator keeps two row blocks of the matrix, its
    slice of the vector and of the result in registers: a spill would send the
    matrix through scratch memory (measured earlier: 5.6 ms instead of 1.1)."""
    table = _resource_table(
        os.path.join(ROOT, "bayes-bridge_amd", "csrc", "dense.hip"), tmp_path)
    fused = {k: v for k, v in table.items() if "dense_fused_kernel" in k}
    assert len(fused) >= 3
    for name, res in fused.items():
        assert res["VGPRs"] <= 128, (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)
