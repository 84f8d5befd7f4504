"""CPU: static wait-state check of the gfx950 ISA around every inline-asm VALU statement (tools/isa_hazard_lint.py).

hipcc's hazard recogniser does not look inside inline asm.  Round 5 met both directions on the GPU before this check existed: an MFMA /
v_permlane16_swap reading a register an asm statement had just written (1e-3 errors at depth), and asm statements reading v_exp_f32 and MFMA
results too early (garbage in one attention instance, stale softmax reference points in the others).  The rules and their numbers are in the
tool's docstring; the files below are the ones whose kernels contain asm VALU statements, compiled with their Makefile flags."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

FILES = [("vit_attn_split.hip", ["-fno-slp-vectorize"]), ("vit_attn.hip", ["-fno-slp-vectorize"]), ("gemm.hip", []),
         ("flash_attn.hip", ["-mllvm", "-amdgpu-mfma-vgpr-form"])]


@pytest.mark.skipif(shutil.which(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")) is None, reason="needs hipcc")
@pytest.mark.parametrize("name,flags", FILES)
def test_no_unprotected_hazard_around_inline_asm(name, flags):
    import isa_hazard_lint as lint
    text = lint.compile_to_asm(os.path.join(ROOT, "hipie_amd", "csrc", name), flags)
    assert text.count(";;#ASMSTART") > 0                    # the marker the check relies on is there
    findings = lint.lint(text, name)
    assert not findings, "\n".join("%s: %s: line %d: %s" % f for f in findings[:10])


def test_lint_flags_the_two_known_patterns():
    """the checker itself: an asm statement right behind the v_exp_f32 / MFMA that writes its operand, and an MFMA right behind an asm write"""
    import isa_hazard_lint as lint
    bad = "\n".join(["k:", "\tv_exp_f32_e32 v1, v0", "\t;;#ASMSTART", "\tv_cvt_pk_f16_f32 v2, v1, v3", "\t;;#ASMEND",
                     "\tv_mfma_f32_32x32x16_f16 v[16:31], v[2:5], v[6:9], v[16:31]", "\t;;#ASMSTART", "\tv_max3_f32 v40, v40, v16, v17", "\t;;#ASMEND"])
    msgs = [f[3] for f in lint.lint(bad, "x")]
    assert any("transcendental" in m for m in msgs) and any("MFMA result" in m for m in msgs) and any("after asm" in m for m in msgs)
    good = "\n".join(["k:", "\tv_exp_f32_e32 v1, v0", "\ts_nop 0", "\t;;#ASMSTART", "\tv_cvt_pk_f16_f32 v2, v1, v3", "\t;;#ASMEND", "\ts_nop 1",
                      "\tv_mfma_f32_32x32x16_f16 v[16:31], v[2:5], v[6:9], v[16:31]", "\ts_nop 10", "\t;;#ASMSTART", "\tv_max3_f32 v40, v40, v16, v17", "\t;;#ASMEND"])
    assert lint.lint(good, "x") == []
