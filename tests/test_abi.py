"""CPU: the C-ABI library builds, loads, and exports every symbol include/hipie_mi355.h declares (no compute calls)."""
import ctypes
import os
import re

from hipie_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "hipie_mi355.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hipie_[a-z_0-9]+)\s*\(", txt)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_all():
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert lib.hipie_version() == 13
    assert lib.hipie_last_error() == b""


def test_argument_validation_without_gpu():
    """bad arguments are rejected on the host before any launch: callable without a device."""
    lib = _lib.load()
    rc = lib.hipie_mask_einsum(None, None, None, 1, 300, 256, 4096, 0, 0, None)
    assert rc == -22 and b"null" in lib.hipie_last_error()
    rc = lib.hipie_dynamic_mask(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16),
                                1, 1, 8, 8, 8, 3, 0, None)
    assert rc == -22 and b"up=3" in lib.hipie_last_error()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from hipie_amd import ops
    v = torch.zeros(1, 4, 2, 2)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.ms_deform_attn_forward(v, torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 2, 1, 1, 2), torch.zeros(1, 1, 2, 1, 1))


def test_new_entry_points_validate_on_the_host():
    lib = _lib.load()
    p = ctypes.c_void_p(256)
    assert lib.hipie_batched_nms(p, p, p, p, p, 1, 2000, 0.7, 1, None) == -22 and b"Q=2000" in lib.hipie_last_error()
    assert lib.hipie_sem_pan(p, p, p, p, p, p, p, p, 10, 16, 200, 8, 8, 4, 32, 32, 32, 32, 0, None) == -22
    assert b"C=200" in lib.hipie_last_error()
    assert lib.hipie_sem_pan(p, p, p, p, p, p, p, p, 10, 10, 5, 8, 8, 4, 32, 32, 32, 32, 0, None) == -22      # Npad % 16
    assert lib.hipie_vit_attn_fused(p, p, p, p, 1, 14, 14, 1, 80, 0.1, 2, None) == -22 and b"64-wide" in lib.hipie_last_error()
    assert lib.hipie_vit_attn_rel(p, p, p, p, 1, 14, 200, 1, 80, 2, 0, None) == -22 and b"wider than 96" in lib.hipie_last_error()
    assert lib.hipie_vit_attn_rel(p, p, p, p, 1, 14, 14, 1, 48, 2, 0, None) == -22 and b"head_dim" in lib.hipie_last_error()
    assert lib.hipie_mask_finalize(p, 0, None, 1, 8, 8, 4, 40, 32, 32, 32, 0.5, p, None) == -22              # crop outside the mask
    assert lib.hipie_add_layernorm_rows(p, None, p, p, None, p, 4, 6, 1e-6, 0, 0, 0, None, None, None) == -22  # C % 4
    # HIPIE_K_HL8_HI (keys = hi halves of an HL8 buffer, which are fp16) with bf16 operands would read garbage keys: refused
    st = [64 * 128, 128, 64] * 4
    assert lib.hipie_flash_attn(p, p, p, p, 1, 2, 64, 64, 64, *st, None, None, 0, 0, None, 1.0, 0.0, 2 | 0x200, None) == -22
    assert b"HL8_HI" in lib.hipie_last_error()
    # hipie_gemm_ln: residual required, K % 32, split operands only, fp32 output rows of at least 256
    assert lib.hipie_gemm_ln(p, 256, p, 512, p, None, 256, p, p, 1e-5, p, 256, p, 512, 10, 256, 0, 1.0, None) == -22 and b"null" in lib.hipie_last_error()
    assert lib.hipie_gemm_ln(p, 256, p, 512, p, p, 256, p, p, 1e-5, p, 256, p, 512, 10, 250, 0, 1.0, None) == -22 and b"K=250" in lib.hipie_last_error()
    assert lib.hipie_gemm_ln(p, 256, p, 512, p, p, 256, p, p, 1e-5, p, 256, p, 512, 10, 256, 1, 1.0, None) == -22 and b"format" in lib.hipie_last_error()
    assert lib.hipie_gemm_ln(p, 256, p, 512, p, p, 256, p, p, 1e-5, p, 200, p, 512, 10, 256, 0, 1.0, None) == -22 and b"stride" in lib.hipie_last_error()
    # empty work is a no-op even with null data pointers
    assert lib.hipie_batched_nms(None, None, None, None, None, 0, 0, 0.7, 1, None) == 0
    assert lib.hipie_mask_finalize(None, 0, None, 0, 8, 8, 4, 32, 32, 32, 32, 0.5, None, None) == 0
