"""ISA lint of the inline-asm MFMA kernels (tests/support/isa_lint.py): the gfx950 code objects of the library as built are disassembled
and every kernel that issues a matrix instruction is checked for (a) scratch traffic between its first and last MFMA and (b) the
hand-written `s_nop` drain sitting between an MFMA and the first access of its accumulator by anything but the next MFMA of the chain,
and (c) two wait states between a VALU write and an MFMA that reads the register (r04 late: the distance hipcc keeps for its builtin).
Two negative controls prove the lint can see what it is for: kernels_posterior.hip rebuilt WITHOUT the drain, and rebuilt without the
empty volatile asms that pin the accumulators behind it (the r03 incident: hipcc then schedules epilogue reads above the drain) --
both must be flagged.  CPU only: hipcc cross-compiles, llvm-objdump disassembles."""
import os
import subprocess

import pytest

from bogp import _lib
from tests.support import isa_lint

CSRC = os.path.join(os.path.dirname(_lib.LIB_PATH), "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(isa_lint.LLVM_BIN, "llvm-objdump"))),
                                reason="needs the ROCm toolchain (hipcc, llvm-objdump)")


@pytest.fixture(scope="module")
def report():
    return isa_lint.lint_library(_lib.LIB_PATH)


def test_every_mfma_kernel_of_the_library_is_found(report):
    names = " ".join(report)
    for family in ("k_contract16", "k_sweep_small", "k_mm128", "k_elim_step", "k_elim_update_b", "k_elim_panel_b", "k_chol_step", "k_tri_gemm", "k_gemm64"):
        assert family in names, "no MFMA kernel of the %s family in the disassembly" % family
    assert len(report) >= 50


def test_no_scratch_in_the_main_loops_and_every_accumulator_access_is_behind_its_drain(report):
    bad = {k: v for k, v in report.items() if v}
    assert not bad, "\n".join("%s:\n    %s" % (k, "\n    ".join(v[:6])) for k, v in bad.items())


def _variant(tmp_path, define):
    obj = str(tmp_path / ("posterior_%s.o" % define))
    subprocess.run([HIPCC, "-D" + define, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I/opt/rocm/include", "-c",
                    os.path.join(CSRC, "kernels_posterior.hip"), "-o", obj], check=True, capture_output=True)  # fmt: skip
    return isa_lint.lint_library(obj, only=("k_contract16",))


def test_a_build_without_the_drain_is_flagged(tmp_path):
    rep = _variant(tmp_path, "BOGP_LINT_NO_DRAIN")
    assert rep and all(v for v in rep.values()), "the lint missed a kernel whose accumulators are read with no drain at all"
    assert all(any(x.startswith("(b)") for x in v) for v in rep.values())


def test_a_build_without_the_accumulator_fences_is_flagged(tmp_path):
    """kernels_posterior.hip without the `asm volatile("" : "+v"(acc))` statements behind the drain: to the compiler an inline-asm
    MFMA's result is ready at once, so it hoists epilogue reads above the `s_nop`s (observed with ROCm 7.2; this is the latent hazard
    r03 found at N = 100) -- the drain is still in the code, yet no longer between the MFMA and the read."""
    rep = _variant(tmp_path, "BOGP_LINT_NO_FENCE")
    flagged = [k for k, v in rep.items() if v]
    assert flagged, "hipcc no longer moves reads above an unfenced drain: re-examine whether the fences are still needed"
    assert any("k_contract16ILi4ELi0" in k for k in flagged)  # the dominant kernel of the C3 sweep among them


def test_the_lint_rules_on_hand_made_sequences():
    mk = lambda rows: [(4 * i, mn, ops) for i, (mn, ops) in enumerate(rows)]  # noqa: E731
    mf = ("v_mfma_f64_16x16x4_f64", "v[0:7], v[20:21], v[22:23], v[0:7]")
    assert isa_lint.lint_function(mk([mf, mf, ("s_nop", "15"), ("s_nop", "2"), ("v_add_f64", "v[30:31], v[0:1], v[2:3]"), ("s_endpgm", "")])) == []
    bad = isa_lint.lint_function(mk([mf, ("s_nop", "15"), ("v_add_f64", "v[30:31], v[0:1], v[2:3]"), ("s_endpgm", "")]))
    assert len(bad) == 4 and all("16 wait states" in b for b in bad)  # v0..v3, 16 < 19
    # another MFMA reading the result as an A operand is an access too; the accumulate chain is not
    assert isa_lint.lint_function(mk([mf, ("v_mfma_f64_16x16x4_f64", "v[8:15], v[0:1], v[22:23], v[8:15]"), ("s_endpgm", "")]))
    # a loop: the read at the top is reached from the MFMA at the bottom through the back edge
    loop = mk([("v_mov_b32_e32", "v40, v3"), mf, ("s_cbranch_scc1", str(65536 - 3)), ("s_endpgm", "")])
    assert any("v_mov_b32" in b for b in isa_lint.lint_function(loop))
    # (c) a VALU result read by an MFMA at once / one state later is flagged, two states later (s_nop 1, or two other instructions) is not
    mv = ("v_mov_b32_e32", "v20, v3")
    assert any(b.startswith("(c)") for b in isa_lint.lint_function(mk([mv, mf, ("s_endpgm", "")])))
    assert any(b.startswith("(c)") for b in isa_lint.lint_function(mk([mv, ("s_nop", "0"), mf, ("s_endpgm", "")])))
    assert isa_lint.lint_function(mk([mv, ("s_nop", "1"), mf, ("s_nop", "15"), ("s_nop", "2"), ("s_endpgm", "")])) == []
    assert isa_lint.lint_function(mk([("v_mul_f64", "v[0:1], v[30:31], v[32:33]"), ("s_waitcnt", "vmcnt(0)"), ("s_nop", "0"), mf, ("s_endpgm", "")])) == []  # hipcc's own spacing
    assert any(b.startswith("(c)") for b in isa_lint.lint_function(mk([("v_mul_f64", "v[6:7], v[30:31], v[32:33]"), ("s_nop", "0"), mf, ("s_endpgm", "")])))  # as SrcC
    # scratch between the first and the last MFMA
    assert any(b.startswith("(a)") for b in isa_lint.lint_function(mk([mf, ("scratch_store_dwordx2", "off, v[50:51], off"), mf, ("s_endpgm", "")])))


def test_instruction_mix_of_the_producers_is_reported_from_the_disassembly():
    """tools/isa_valu_mix.py: the per-pair FP64 VALU count of both correlation producers read off the shipped library --
    kernel A' must keep the distance off the VALU (no more than 45 DP instructions a pair for Matern-5/2, kernel A: 39 + 40 at d = 20)."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.LIB_PATH)))
    spec = importlib.util.spec_from_file_location("isa_valu_mix", os.path.join(root, "tools", "isa_valu_mix.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    seen = {}
    for text in isa_lint.disassemble_library(_lib.LIB_PATH):
        for name, insns in mod.kernels(text).items():
            for tag in ("k_corr_mfmaILi3E", "k_corr_chunkILi3ELi0E"):
                if tag in name:
                    blocks = [mod.classify(b) for b in mod.basic_blocks(insns)]
                    prof = max(blocks, key=lambda c: (c["stores"], c["dp_valu"]))
                    seen[tag] = (prof["dp_valu"] / prof["stores"], prof["stores"], max(c["mfma"] for c in blocks))
    assert set(seen) == {"k_corr_mfmaILi3E", "k_corr_chunkILi3ELi0E"}
    assert seen["k_corr_mfmaILi3E"][1] == 16 and seen["k_corr_mfmaILi3E"][0] <= 45.0 and seen["k_corr_mfmaILi3E"][2] >= 4
    assert seen["k_corr_chunkILi3ELi0E"][1] == 8 and seen["k_corr_chunkILi3ELi0E"][2] == 0
