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


# (VALS, WIDE, KP, FOLD, DENSEP, PACK) of every tiled_spmv_kernel instantiation
TILED_INSTANCES = [("0", "0", "0", "0", "0", "0"), ("0", "1", "0", "0", "0", "0"),
                   ("1", "0", "0", "0", "0", "0"), ("1", "1", "0", "0", "0", "0"),
                   ("0", "1", "1", "0", "0", "0"), ("0", "1", "2", "0", "0", "0"),
                   ("1", "1", "1", "0", "0", "0"), ("0", "1", "0", "1", "0", "0"),
                   ("0", "1", "0", "0", "1", "0"),
                   # value-free, one right-hand side, ids in groups of five
                   ("0", "0", "0", "0", "0", "1"), ("0", "1", "0", "0", "0", "1"),
                   ("0", "1", "0", "1", "0", "1"), ("0", "1", "0", "0", "1", "1")]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_tiled_kernels_do_not_spill(tmp_path):
    table = _resource_table(
        os.path.join(ROOT, "bayes-bridge_amd", "csrc", "spmv_tiled.hip"),
        tmp_path)
    tiled = {k: v for k, v in table.items() if "tiled_spmv_kernel" in k}
    # exactly the instantiations build_tiled sets attributes on:
    # (VALS, WIDE, KP, FOLD, DENSEP, PACK) -- value-free and valued, 8- and
    # 16-byte slice refills, the K-column ones (KP = 1: two chains, KP = 2:
    # four), the one that carries the CG direction step, the one with a mixed
    # design's dense block in its epilogue, and the value-free single-chain
    # ones again for ids packed in groups of five; several sit at the 128-VGPR
    # budget: a dropped or renamed one must be noticed
    got = sorted(re.search(
        r"tiled_spmv_kernelILb(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)E",
        k).groups()
        for k in tiled)
    assert got == sorted(TILED_INSTANCES), got
    for name, res in tiled.items():
        assert res["VGPRs"] <= 128, (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["SGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)
    # a mixed design's dense block in one pass (wave per row: 1 / 2 / 4 / 8
    # column pairs per lane; workgroup per row block: 1 / 2 / 4 pairs per
    # thread, up to 8192 columns): rows of D, the slice of v_D and of D^T t in
    # registers -- a spill would stream the block through scratch
    dense = {k: v for k, v in table.items() if "hyb_dense_fused" in k}
    assert len(dense) == 7, sorted(dense)
    for name, res in dense.items():
        # (the eight-pair form runs 512 threads: twice the registers per lane)
        assert res["VGPRs"] <= (256 if "ELi512EE" in name else 128), (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_dense_fused_kernels_fit_the_register_budget(tmp_path):
    """The single-pass dense operator keeps two row blocks of the matrix, its
    slice of the vector and of the result in registers: a spill would send the
    matrix through scratch memory (measured earlier: 5.6 ms instead of 1.1)."""
    table = _resource_table(
        os.path.join(ROOT, "bayes-bridge_amd", "csrc", "dense.hip"), tmp_path)
    fused = {k: v for k, v in table.items() if "dense_fused" in k}
    # f32: register form and LDS-DMA ring, one and two column groups; f64: the
    # pair-layout kernels, two and four pairs per thread
    assert len(fused) == 8, sorted(fused)
    for name, res in fused.items():
        assert res["VGPRs"] <= 128, (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)


DENSE_BATCH_SRC = os.path.join(ROOT, "bayes-bridge_amd", "csrc",
                               "dense_batch.hip")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_dense_batch_kernels_keep_their_rings_in_architectural_vgprs(tmp_path):
    """dense_{dot,tdot}_kd_kernel<T, NG>: accumulators AND the asm-issued
    register ring live in architectural VGPRs (accumulators in AGPRs halve the
    f64 MFMA's issue rate, profiles/r03_mfma_f64_acc.txt; a ring register that
    the compiler copies or spills while its asm load is in flight is silent
    corruption).  So: <= 256 VGPRs, no AGPRs at all, no scratch, no spills --
    for each of the eight instantiations."""
    table = _resource_table(DENSE_BATCH_SRC, tmp_path)
    kd = {k: v for k, v in table.items() if "_kd_kernel" in k}
    got = sorted(re.search(r"dense_(t?dot)_kd_kernelI([fd])Li(\d)E", k).groups()
                 for k in kd)
    assert got == sorted((o, t, g) for o in ("dot", "tdot") for t in "fd"
                         for g in "12"), got
    for name, res in kd.items():
        assert res["VGPRs"] <= 256, (name, res)
        assert res["AGPRs"] == 0, (name, res)
        assert res["VGPRs Spill"] == 0, (name, res)
        assert res["SGPRs Spill"] == 0, (name, res)
        assert res["ScratchSize [bytes/lane]"] == 0, (name, res)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_asm_mfmas_keep_their_wait_states(tmp_path):
    """Disassembly check (llvm-objdump of the gfx950 code object, CPU only):
    every v_mfma_f64_16x16x4_f64 of dense_batch.hip has no VALU write of its
    operands within 2 wait states in front of it and no access to its result
    within 18 behind it other than the accumulate chain (tests/asm_hazards.py).
    The MFMAs are issued from inline asm, which the compiler's hazard
    recogniser does not see: a compiler update could otherwise break the
    kernels silently for every shape the numeric tests do not run."""
    import asm_hazards
    dis = asm_hazards.disassemble(DENSE_BATCH_SRC, tmp_path)
    bad, n_mfma = asm_hazards.check(dis, "_kd_kernel")
    assert n_mfma >= 8 * 16          # every instantiation was looked at
    assert not bad, "\n".join(bad[:10])
    # the checker itself: the documented first version (conversion directly in
    # front of its MFMA) and a result read after 4 wait states must both fail
    neg = asm_hazards.disassemble(
        os.path.join(ROOT, "tests", "fixtures", "mfma_hazard_negative.hip"),
        tmp_path)
    bad_cvt, n1 = asm_hazards.check(neg, "cvt_in_front_of_its_mfma")
    bad_read, n2 = asm_hazards.check(neg, "result_read_too_early")
    assert n1 == 1 and n2 == 1
    assert len(bad_cvt) == 1 and "VALU write 0 wait state" in bad_cvt[0]
    assert len(bad_read) == 1 and "touched after 4 wait state" in bad_read[0]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_registers_of_asm_issued_loads_are_untouched_while_in_flight(tmp_path):
    """The stream loads of spmv_tiled.hip and dense_batch.hip are issued from
    inline asm into register rings and retired with hand-counted `s_waitcnt
    vmcnt(N)`; the compiler does not know those registers are busy.  Walk the
    disassembly forward from every vector-memory load: nothing may name its
    destination before a wait that covers it (tests/asm_hazards.py rule C;
    strict for the dense batch kernels, copy/spill/hoist detection for the
    tiled kernel -- see check_inflight)."""
    import asm_hazards
    csrc = os.path.join(ROOT, "bayes-bridge_amd", "csrc")
    dis = asm_hazards.disassemble(DENSE_BATCH_SRC, tmp_path)
    bad, n = asm_hazards.check_rings(dis, "_kd_kernel", strict=True)
    assert n >= 8 * 40 and not bad, "\n".join(bad[:6])
    dis = asm_hazards.disassemble(os.path.join(csrc, "spmv_tiled.hip"),
                                  tmp_path)
    bad, n = asm_hazards.check_rings(dis, "tiled_spmv_kernel", strict=False)
    assert n >= len(TILED_INSTANCES) * 20 and not bad, "\n".join(bad[:6])
    neg = asm_hazards.disassemble(
        os.path.join(ROOT, "tests", "fixtures", "mfma_hazard_negative.hip"),
        tmp_path)
    for strict in (True, False):
        bad, n = asm_hazards.check_rings(
            neg, "ring_register_touched_in_flight", strict=strict)
        assert n == 1 and len(bad) == 1 and "in-flight load" in bad[0]
    # copies of ring registers (what broke the early re-arm of the tiled
    # kernel's slots in round 4: asm_hazards.check_ring_copies)
    bad, n = asm_hazards.check_ring_copies(dis, "tiled_spmv_kernel")
    assert n >= len(TILED_INSTANCES) * 20 and not bad, "\n".join(bad[:6])
    bad, n = asm_hazards.check_ring_copies(
        neg, "ring_register_copied_before_its_wait")
    assert n == 1 and len(bad) == 1 and "v_mov_b32" in bad[0]
