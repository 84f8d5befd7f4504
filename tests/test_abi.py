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
    assert lib.hipie_version() == 1
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
