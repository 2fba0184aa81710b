"""ISA lint for the inline-asm MFMA kernels of libbogp (gfx950).

The FP64 matrix instructions are issued from inline asm (`v_mfma_f64_16x16x4_f64` / `v_mfma_f64_4x4x4_4b_f64`, csrc/kernels_posterior.hip,
kernels_small.hip, kernels_chol.hip, kernels_gemm.hip, kernels_point.hip): the compiler neither sees that an accumulator is still in
flight when the asm statement ends nor pads the hazard -- the sources drain by hand (`s_nop` runs) and pin the accumulators behind the
drain with empty volatile asms.  A compiler bump can silently undo either.  This module checks the EMITTED code instead of the
convention:

  (a) no scratch (spill) access between the first and the last matrix instruction of a kernel -- a spilled accumulator or operand
      ring inside the main loop is a 2-10x slowdown that no numerical test notices;
  (b) every access to a register an MFMA has written -- by anything but an MFMA that takes the whole range as its accumulator
      operand (the accumulate chain) -- has at least `REQUIRED_WAIT_STATES` wait states of `s_nop` (`s_nop N` = N + 1) between that MFMA
      and itself.  Only the nops count: they are the drain the source wrote; any other instruction in between is there by the
      scheduler's choice and may be gone with the next compiler.  19 is what hipcc (ROCm 7.2) itself places behind the BUILTIN form of
      v_mfma_f64_16x16x4_f64 on gfx950 before a VALU reads the result (`s_nop 15; s_nop 2`: the 16-pass DGEMM write -> VALU / VMEM / LDS
      read rule of LLVM's GCNHazardRecognizer), the largest of the family, applied to every MFMA here;
  (c) a register written by a VALU instruction is not read by an MFMA (as A, B or C) with fewer than `VALU_TO_MFMA_WAIT_STATES` = 2
      wait states in between -- what hipcc places between a `v_mov` / `v_mul_f64` and the builtin MFMA that reads its result (`s_nop 1`).
      Any instruction counts as one state here, `s_nop N` as N + 1.  The sources keep this distance by hand where they build MFMA operands
      with VALU code (`s_nop 7; s_nop 7` in front of the chains of mma_64 / elim_step_block): to the compiler an inline-asm MFMA is just
      another instruction, it pads nothing.

The scan is linear in address order and every taken branch edge is followed from its target while a result is still pending (an MFMA at
the bottom of a loop against a read at its top; an epilogue placed before the loop in the address space).  A pass is a guarantee of
the convention; a failure may be a false alarm to be looked at -- none exists in the shipped build.

Test infrastructure only (tests/test_isa_lint.py); runs on the CPU: hipcc cross-compiles and llvm-objdump disassembles without a GPU."""
import os
import re
import shutil
import subprocess
import tempfile

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
REQUIRED_WAIT_STATES = 19
VALU_TO_MFMA_WAIT_STATES = 2
_REG = re.compile(r"\b([va])(?:\[(\d+):(\d+)\]|(\d+)\b)")
_FUNC = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
_INSN = re.compile(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):")


def disassemble_code_object(path):
    return subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "-d", "--mcpu=gfx950", path], check=True, capture_output=True, text=True).stdout


def disassemble_library(lib_path):
    """Disassembly text of every gfx950 code object bundled in a host library / object file."""
    work = tempfile.mkdtemp(prefix="bogp_isa_")
    try:
        local = os.path.join(work, os.path.basename(lib_path))
        shutil.copy(lib_path, local)  # llvm-objdump --offloading writes the bundles next to its input
        subprocess.run([os.path.join(LLVM_BIN, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=work)
        out = []
        for f in sorted(os.listdir(work)):
            if "gfx950" in f and "amdgcn" in f:
                out.append(disassemble_code_object(os.path.join(work, f)))
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def _regs(text):
    """Set of (file, index) registers named in an operand string."""
    s = set()
    for m in _REG.finditer(text):
        if m.group(2) is not None:
            s.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            s.add((m.group(1), int(m.group(4))))
    return s


def _split_operands(ops):
    out, depth, cur = [], 0, ""
    for ch in ops:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_functions(asm_text):
    """{mangled name: [(address, mnemonic, operand string)]}"""
    funcs, cur = {}, None
    for line in asm_text.splitlines():
        m = _FUNC.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = _INSN.match(line)
        if m:
            cur.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return funcs


def lint_function(insns, required=REQUIRED_WAIT_STATES, nops_only=True):
    """Violations of rules (a), (b) and (c) in one kernel: a list of strings (empty = clean).  `insns` as parse_functions gives them."""
    mfma_idx = [i for i, (_, mn, _) in enumerate(insns) if mn.startswith("v_mfma")]
    if not mfma_idx:
        return []
    bad = []
    first, last = mfma_idx[0], mfma_idx[-1]
    n_scratch = sum(1 for _, mn, _ in insns[first : last + 1] if mn.startswith("scratch_"))
    if n_scratch:
        bad.append("(a) %d scratch accesses between the first and the last MFMA" % n_scratch)
    addr_index = {a: i for i, (a, _, _) in enumerate(insns)}
    n = len(insns)

    # successors of every instruction
    succ = [[] for _ in range(n)]
    for i, (addr, mn, ops) in enumerate(insns):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            try:  # the operand is the signed 16-bit offset in dwords from the NEXT instruction
                imm = int(ops.split()[0], 0)
                tgt = addr_index.get(addr + 4 + 4 * (imm - 65536 if imm >= 32768 else imm))
            except (ValueError, IndexError):
                tgt = None
            if tgt is not None:
                succ[i].append(tgt)
        if mn not in ("s_branch", "s_endpgm") and i + 1 < n:
            succ[i].append(i + 1)

    # Forward data-flow to a fixed point.  State before an instruction: {register: wait states of s_nop seen since the MFMA that wrote
    # it}, the MINIMUM over all paths; a register leaves the state once `required` states have passed (drained) or it is accessed.
    state_in = [None] * n
    state_in[0] = {}
    work = [0]
    flagged = {}
    while work:
        i = work.pop()
        st = dict(state_in[i])
        addr, mn, ops = insns[i]
        if mn.startswith("v_mfma"):
            o = _split_operands(ops)
            dst, a_b, c = _regs(o[0]), _regs(o[1]) | _regs(o[2]), (_regs(o[3]) if len(o) > 3 else set())
            chain = c == dst  # the accumulate chain: D of the previous MFMA taken whole as C
            for r in a_b | (set() if chain else c | dst):
                if r in st:
                    flagged[(addr, r)] = "(b) %s at 0x%x uses %s%d with %d wait states of s_nop after the MFMA that wrote it (< %d)" % (mn, addr, r[0], r[1], st[r], required)
                    st.pop(r)
            if not nops_only:
                st = {r: w + 1 for r, w in st.items() if w + 1 < required}
            for r in dst:
                st[r] = 0
        elif st:
            for r in _regs(ops):
                if r in st:
                    flagged[(addr, r)] = "(b) %s at 0x%x uses %s%d with %d wait states of s_nop after the MFMA that wrote it (< %d)" % (mn, addr, r[0], r[1], st[r], required)
                    st.pop(r)
            step = 0 if nops_only else 1
            if mn == "s_nop":
                try:
                    step = int(ops.split()[0], 0) + 1
                except (ValueError, IndexError):
                    step = 1
            if step:
                st = {r: w + step for r, w in st.items() if w + step < required}
        for j in succ[i]:
            old = state_in[j]
            if old is None:
                state_in[j] = dict(st)
                work.append(j)
            else:
                changed = False
                for r, w in st.items():
                    if r not in old or w < old[r]:
                        old[r] = w
                        changed = True
                if changed:
                    work.append(j)
    bad.extend(sorted(set(flagged.values())))

    # (c): walk back from every MFMA over at most VALU_TO_MFMA_WAIT_STATES - 1 wait states of predecessors
    pred = [[] for _ in range(n)]
    for i in range(n):
        for j in succ[i]:
            pred[j].append(i)

    def states(j):
        _, mn, ops = insns[j]
        if mn == "s_nop":
            try:
                return int(ops.split()[0], 0) + 1
            except (ValueError, IndexError):
                return 1
        return 1

    close = set()
    for i in mfma_idx:
        addr, mn, ops = insns[i]
        o = _split_operands(ops)
        src = _regs(o[1]) | _regs(o[2]) | (_regs(o[3]) if len(o) > 3 else set())
        stack = [(j, 0) for j in pred[i]]  # (instruction, wait states between it and the MFMA)
        seen = set()
        while stack:
            j, gap = stack.pop()
            if (j, gap) in seen or gap >= VALU_TO_MFMA_WAIT_STATES:
                continue
            seen.add((j, gap))
            _, pmn, pops = insns[j]
            if pmn.startswith("v_") and not pmn.startswith("v_mfma") and pmn != "v_nop":
                po = _split_operands(pops)
                hit = (_regs(po[0]) if po else set()) & src
                if hit:
                    r = sorted(hit)[0]
                    close.add("(c) %s at 0x%x reads %s%d %d wait state(s) after %s wrote it (< %d)" % (mn, addr, r[0], r[1], gap, pmn, VALU_TO_MFMA_WAIT_STATES))
            for q in pred[j]:
                stack.append((q, gap + states(j)))
    bad.extend(sorted(close))
    return bad


def lint_library(lib_path, only=None):
    """{kernel: [violations]} over every MFMA kernel of the library (or those whose mangled name contains one of `only`)."""
    report = {}
    for text in disassemble_library(lib_path):
        for name, insns in parse_functions(text).items():
            if only and not any(o in name for o in only):
                continue
            if any(mn.startswith("v_mfma") for _, mn, _ in insns):
                report[name] = lint_function(insns)
    return report
