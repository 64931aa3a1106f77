"""Static checks on the gfx950 disassembly of kernels that issue MFMAs from
inline asm (csrc/dense_batch.hip).  TEST INFRASTRUCTURE.

An MFMA written in `asm volatile` is invisible to hipcc's hazard recogniser:
nothing pads the wait states between a VALU write and the matrix core's read of
that register, or between an MFMA and the first non-accumulating access to its
result (cdna_hip_programming.md 5.7 item 2; dense_batch.hip:111-127 describes
the first version of the kernels, which read stale operands exactly that way).
The rules checked here, for every `v_mfma_f64_16x16x4_f64 D, A, B, C`:

  A  no VALU instruction that writes a register of A, B or C within the
     VALU_TO_MFMA = 2 wait states in front of it (`s_nop 1`; LLVM
     GCNHazardRecognizer "LegacyVALUWritesVGPRWaitStates" for DGEMM);
  B  within MFMA_TO_USE = 18 wait states behind it no instruction touches a
     register of D -- except an MFMA that accumulates into exactly D (C == D,
     the accumulate chain: 0 wait states).  18 is the largest entry of the
     DGEMM 16x16 rows (result read by VMEM / export); a VALU read needs 11.

Wait states are counted as LLVM does: one per instruction issued, N + 1 for
`s_nop N`.  Control flow is followed: the walk goes through every predecessor
(fall-through and branches to the instruction) resp. successor, so a hazard
that only exists around a loop back-edge is seen.
"""
import os
import re
import subprocess

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
MFMA = "v_mfma_f64_16x16x4_f64"
VALU_TO_MFMA = 2
MFMA_TO_USE = 18


def disassemble(src, workdir, extra_flags=()):
    """hipcc (device only) -> unbundle -> llvm-objdump -d.  Returns the text."""
    co = os.path.join(str(workdir), "k.co")
    elf = os.path.join(str(workdir), "k.elf")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950",
                    "--offload-device-only", *extra_flags, "-c", src, "-o", co],
                   check=True, capture_output=True, timeout=900)
    subprocess.run([os.path.join(LLVM_BIN, "clang-offload-bundler"),
                    "--unbundle", "--type=o", "--input=" + co,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--output=" + elf], check=True, capture_output=True)
    out = subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", elf],
                         check=True, capture_output=True, text=True)
    return out.stdout


class Instr:
    __slots__ = ("addr", "mnem", "ops", "target", "text")

    def __init__(self, addr, mnem, ops, target, text):
        self.addr, self.mnem, self.ops = addr, mnem, ops
        self.target, self.text = target, text


_FUNC = re.compile(r"^([0-9a-f]+) <(\S+)>:")
_LINE = re.compile(r"^\s+(\S+)\s*(.*?)\s*// ([0-9A-Fa-f]+): [0-9A-Fa-f ]+?"
                   r"(?:<(\S+?)\+0x([0-9a-f]+)>|<(\S+?)>)?\s*$")


def parse(dis):
    """{function: [Instr]} from llvm-objdump -d output."""
    funcs, cur, starts = {}, None, {}
    for line in dis.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = m.group(2)
            funcs[cur] = []
            starts[cur] = int(m.group(1), 16)
            continue
        if cur is None:
            continue
        m = _LINE.match(line)
        if not m:
            continue
        mnem, rest, addr = m.group(1), m.group(2), int(m.group(3), 16)
        target = None
        if mnem.startswith("s_branch") or mnem.startswith("s_cbranch"):
            if m.group(4):
                target = starts.get(m.group(4), 0) + int(m.group(5), 16)
            elif m.group(6):
                target = starts.get(m.group(6))
            ops = []
        else:
            ops = [o.strip() for o in _split_ops(rest)]
        funcs[cur].append(Instr(addr, mnem, ops, target, line.strip()))
    return funcs


def _split_ops(rest):
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


_REG = re.compile(r"^(v|a)(?:(\d+)|\[(\d+):(\d+)\])$")


def regs(op):
    """Set of ('v'|'a', index) an operand names (modifiers like neg() or
    op_sel suffixes stripped); empty for SGPRs, literals, labels."""
    op = op.strip()
    m = re.match(r"^(?:-|\||neg\(|abs\()*\s*((?:v|a)(?:\d+|\[\d+:\d+\]))", op)
    if not m:
        return set()
    m = _REG.match(m.group(1))
    if not m:
        return set()
    if m.group(2) is not None:
        return {(m.group(1), int(m.group(2)))}
    return {(m.group(1), k) for k in range(int(m.group(3)), int(m.group(4)) + 1)}


def wait_states(ins):
    if ins.mnem == "s_nop":
        return int(ins.ops[0], 0) + 1 if ins.ops else 1
    return 1


def is_valu(ins):
    return ins.mnem.startswith("v_") and "mfma" not in ins.mnem \
        and ins.mnem not in ("v_nop",)


def valu_writes(ins):
    """VGPRs a VALU instruction writes."""
    if not ins.ops:
        return set()
    w = regs(ins.ops[0])
    if "swap" in ins.mnem and len(ins.ops) > 1:
        w |= regs(ins.ops[1])
    return w


def all_regs(ins):
    out = set()
    for o in ins.ops:
        out |= regs(o)
    return out


def _graph(code):
    by_addr = {ins.addr: i for i, ins in enumerate(code)}
    succ = [[] for _ in code]
    pred = [[] for _ in code]
    for i, ins in enumerate(code):
        ends = ins.mnem in ("s_endpgm", "s_branch", "s_setpc_b64")
        if not ends and i + 1 < len(code):
            succ[i].append(i + 1)
        if ins.target is not None and ins.target in by_addr:
            succ[i].append(by_addr[ins.target])
    for i, ss in enumerate(succ):
        for s in ss:
            pred[s].append(i)
    return succ, pred


def check_function(code, name=""):
    """List of violation strings for one function's instruction list."""
    succ, pred = _graph(code)
    bad = []
    for i, ins in enumerate(code):
        if ins.mnem != MFMA or len(ins.ops) < 4:
            continue
        d, srcs = regs(ins.ops[0]), regs(ins.ops[1]) | regs(ins.ops[2]) | \
            regs(ins.ops[3])
        # rule A: walk back through every predecessor path
        seen = {}
        stack = [(p, 0) for p in pred[i]]
        while stack:
            j, ws = stack.pop()
            if ws >= VALU_TO_MFMA or seen.get(j, 99) <= ws:
                continue
            seen[j] = ws
            pj = code[j]
            if is_valu(pj) and (valu_writes(pj) & srcs):
                bad.append("%s: VALU write %d wait state(s) in front of the "
                           "MFMA that reads it:\n    %s\n    %s"
                           % (name, ws, pj.text, ins.text))
            for p in pred[j]:
                stack.append((p, ws + wait_states(pj)))
        # rule B: walk forward
        seen = {}
        stack = [(s, 0) for s in succ[i]]
        while stack:
            j, ws = stack.pop()
            if ws >= MFMA_TO_USE or seen.get(j, 99) <= ws:
                continue
            seen[j] = ws
            nj = code[j]
            touched = all_regs(nj) & d
            if touched:
                chain = (nj.mnem == MFMA and len(nj.ops) >= 4
                         and regs(nj.ops[0]) == d and regs(nj.ops[3]) == d
                         and not ((regs(nj.ops[1]) | regs(nj.ops[2])) & d))
                if chain:
                    continue        # D is rewritten: later accesses are its
                bad.append("%s: MFMA result touched after %d wait state(s) "
                           "(< %d):\n    %s\n    %s"
                           % (name, ws, MFMA_TO_USE, ins.text, nj.text))
                continue
            for s in succ[j]:
                stack.append((s, ws + wait_states(nj)))
    return bad


def check(dis, only=None):
    """(violations, number of MFMAs looked at) over every function of a
    disassembly whose name contains `only` (all when None)."""
    bad, n = [], 0
    for name, code in parse(dis).items():
        if only and only not in name:
            continue
        n += sum(1 for ins in code if ins.mnem == MFMA)
        bad += check_function(code, name)
    return bad, n


# ---------------------------------------------------------------------------
# Registers of in-flight loads.  The tiled SpMV kernel (csrc/spmv_tiled.hip)
# and the dense batch kernels issue their stream loads from inline asm into a
# ring of registers and retire them with hand-counted `s_waitcnt vmcnt(N)`.
# The compiler neither counts those loads nor knows that their destinations
# are busy: a copy, spill or re-use of such a register between issue and wait
# would be silent corruption (spmv_tiled.hip:342-349 describes one).
#
# Rule C, for every vector-memory load with a VGPR destination D: on every
# path forward, no instruction names a register of D until an
# `s_waitcnt vmcnt(N)` with N <= (vector-memory instructions issued after the
# load) has been passed -- loads return in order, so at that wait the load has
# landed.  Holds trivially for compiler-issued loads; for the asm rings it
# verifies the counted waits against the actual instruction stream.  (A younger
# LOAD into the same destination is allowed: in-order return makes it land
# last, and its own walk covers what follows.)
_VMEM = re.compile(r"^(global|buffer|flat|scratch)_(load|store|atomic)")
_VMCNT = re.compile(r"vmcnt\((\d+)\)")


def is_vmem(ins):
    return bool(_VMEM.match(ins.mnem))


def vmem_load_dest(ins):
    if not _VMEM.match(ins.mnem) or "_load" not in ins.mnem \
            or "_lds_" in ins.mnem or not ins.ops:
        return set()
    return regs(ins.ops[0])


def check_inflight(code, name="", strict=True):
    """Rule C over one function.  The walk is path-INsensitive.  strict: every
    touch reached without a covering wait is reported (the dense batch kernels'
    loops pass this).  The tiled kernel's stream loop carries a `done` flag the
    walk cannot correlate ("slot 0 hit the end marker, slot 1 consumed" is a
    path of the graph, not of the program), so for it strict=False reports a
    touch only if the touching instruction is not ALSO reachable from the load
    behind a covering wait: a compiler copy or spill of a ring register at loop
    entry (the hazard spmv_tiled.hip:342-349 documents) or a consumer hoisted
    above its wait leave no covered path and are reported; a wrong wait COUNT
    is not -- that one is constexpr arithmetic in the source
    ((RING - 1) * LOADS_PER_STEP), which a compiler update cannot change, and
    the kernel == CPU-emulator bitwise tests cover it on the GPU."""
    succ, _ = _graph(code)
    bad = []
    for i, ins in enumerate(code):
        d = vmem_load_dest(ins)
        if not d:
            continue

        def supersedes(nj):
            addr_regs = set()
            for o in nj.ops[1:]:
                addr_regs |= regs(o)
            return bool(vmem_load_dest(nj) & d) and not (addr_regs & d)
        seen, covered_at, touches = set(), [], []
        stack = [(s, 0) for s in succ[i]]
        while stack:
            j, younger = stack.pop()
            if (j, younger) in seen:
                continue
            seen.add((j, younger))
            nj = code[j]
            if nj.mnem == "s_waitcnt":
                m = _VMCNT.search(" ".join(nj.ops))
                if m and younger >= int(m.group(1)):
                    covered_at.append(j)          # landed on this path
                    continue
            elif all_regs(nj) & d:
                # a YOUNGER load into the same registers (the ring slot
                # re-issued behind a marker batch): loads return in order, the
                # younger one lands last and is tracked on its own
                if not supersedes(nj):
                    touches.append((j, younger))
                continue
            if nj.mnem == "s_endpgm":
                continue
            y2 = min(younger + (1 if is_vmem(nj) else 0), 64)
            for s in succ[j]:
                stack.append((s, y2))
        if not touches:
            continue
        # instructions reachable behind a covering wait, up to the re-issue
        reach = set()
        stack = [] if strict else [s for c in covered_at for s in succ[c]]
        while stack:
            j = stack.pop()
            if j in reach:
                continue
            reach.add(j)
            if j == i or supersedes(code[j]) or code[j].mnem == "s_endpgm":
                continue
            stack.extend(succ[j])
        for j, younger in touches:
            if j not in reach:
                bad.append("%s: register of an in-flight load touched (%d "
                           "younger vector-memory ops, no covering wait):\n"
                           "    %s\n    %s"
                           % (name, younger, ins.text, code[j].text))
    return bad


def check_rings(dis, only=None, strict=True):
    """(violations, loads looked at) of rule C over the matching functions."""
    bad, n = [], 0
    for name, code in parse(dis).items():
        if only and only not in name:
            continue
        n += sum(1 for ins in code if vmem_load_dest(ins))
        bad += check_inflight(code, name, strict)
    return bad, n


# Rule C': copies.  The lenient form of rule C accepts a touch that is also
# reachable behind a covering wait, which let this one through in round 4: a
# row-id value kept live across its ring slot's re-issue made the register
# allocator copy the slot's wait operand (tied in/out) in front of the wait --
# `v_mov_b32 v38, v54` one instruction above `s_waitcnt vmcnt(4)`, v54 still
# the destination of a load in flight: stale row ids, CG iteration counts x 2.5.
# A COPY of a ring register is never one of the kernel's own consumers (those
# are the decode operations behind the wait), so every copy-like instruction
# that the strict, path-insensitive walk reaches without a covering wait is
# reported: no such instruction exists in a correct build, on any path.
_COPY_LIKE = ("v_mov_b32", "v_mov_b64", "v_pk_mov_b32", "v_swap_b32",
              "v_accvgpr_write_b32", "v_writelane_b32", "scratch_store",
              "buffer_store")


def check_ring_copies(dis, only=None):
    """(violations, loads looked at): copy-like instructions reading the
    destination of an in-flight load, reached without a covering wait."""
    bad, n = check_rings(dis, only, strict=True)
    seen, out = set(), []
    for b in bad:
        lines = b.split("\n")
        touch = lines[-1].strip()
        if not touch.startswith(_COPY_LIKE):
            continue
        key = (lines[0].split(":")[0], touch)
        if key not in seen:
            seen.add(key)
            out.append(b)
    return out, n
