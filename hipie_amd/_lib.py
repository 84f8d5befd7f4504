"""ctypes binding of libhipie_mi355.so (the C ABI of include/hipie_mi355.h).

The library is built in-tree (``make -C hipie_amd/csrc`` or ``__graft_entry__.build()``).  There is NO fallback: if the
shared object is missing or does not export a symbol, importing the op layer raises -- the product never silently runs a
PyTorch/CPU substitute for a hand-written kernel.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPIE_LIB_PATH: another build of the SAME ABI, for same-box A/B timing of a kernel change (tools/); the default is the in-tree library
LIB_PATH = os.environ.get("HIPIE_LIB_PATH") or os.path.join(_HERE, "csrc", "libhipie_mi355.so")

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_f = ctypes.c_float

# name -> argtypes, in the order of include/hipie_mi355.h
SIGNATURES = {
    "hipie_version": [],
    "hipie_last_error": [],
    "hipie_msda_forward": [c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 8 + [c_p],
    "hipie_msda_fused_forward": [c_p, c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 10 + [c_l, c_l, c_p],
    "hipie_msda_fused_forward_strided": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 10 + [c_l, c_l, c_p],
    "hipie_flash_attn": [c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_l] * 12 + [c_p, c_p, c_i, c_i, c_p, c_f, c_f, c_i, c_p],
    "hipie_vit_attn": [c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_f, c_i, c_p],
    "hipie_vit_attn_fused": [c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_f, c_i, c_p],
    "hipie_vit_attn_rel": [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p],
    "hipie_bi_xattn": [c_p, c_p, c_p, c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_f, c_i, c_p],
    "hipie_bi_xattn_ws": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l] + [c_i] * 5 + [c_f, c_i, c_p],
    "hipie_bi_xattn_workspace": [c_i] * 5,
    "hipie_mask_einsum": [c_p, c_p, c_p] + [c_i] * 6 + [c_p],
    "hipie_mask_einsum_workspace": [c_i, c_i, c_i],
    "hipie_mask_einsum_ws": [c_p, c_p, c_p, c_p, c_p, c_l] + [c_i] * 6 + [c_p],
    "hipie_mask_einsum16": [c_p, c_p, c_p, c_p, c_p] + [c_i] * 6 + [c_p],
    "hipie_dynamic_mask": [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p],
    "hipie_dynamic_mask16": [c_p, c_p, c_p, c_p] + [c_i] * 7 + [c_p],
    "hipie_dynamic_mask_backward": [c_p] * 7 + [c_i] * 6 + [c_p],
    "hipie_vit_relpos": [c_p, c_p, c_p, c_p, c_p] + [c_i] * 6 + [c_p],
    "hipie_add_layernorm": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_i, c_i, c_p],
    "hipie_add_layernorm_sum": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_i, c_i, c_p],
    "hipie_add_layernorm_rows": [c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_i, c_i, c_p, c_p, c_p],
    "hipie_batched_nms": [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_f, c_i, c_p],
    "hipie_mask_finalize": [c_p, c_i, c_p] + [c_i] * 8 + [c_f, c_p, c_p],
    "hipie_sem_pan": [c_p] * 8 + [c_i] * 11 + [c_p],
    "hipie_sine_embed": [c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_f, c_i, c_p],
    "hipie_box_refine": [c_p, c_p, c_p, c_l, c_f, c_i, c_p],
    "hipie_ref_point_mlp": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_p],
    "hipie_box_head": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_f, c_p],
    "hipie_add_layernorm_dec": [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_f, c_i, c_i, c_p],
    "hipie_add_cast": [c_p, c_p, c_p, c_l, c_i, c_p],
    "hipie_group_norm": [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_i, c_p],
    "hipie_gemm": [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p] + [c_i] * 6 + [c_f, c_f, c_p],
    "hipie_gemm_ln": [c_p, c_l, c_p, c_l, c_p, c_p, c_l, c_p, c_p, c_f, c_p, c_l, c_p, c_l, c_i, c_i, c_i, c_f, c_p],
    "hipie_gemm_gather": [c_p, c_l, c_l, c_p, c_p, c_l, c_p, c_p, c_l, c_p, c_l, c_p] + [c_i] * 6 + [c_f, c_f, c_p],
    "hipie_vit_attn_split": [c_p, c_p, c_p, c_p] + [c_i] * 5 + [c_p],
    "hipie_msda_backward": [c_p] * 9 + [c_i] * 8 + [c_p],
    "hipie_msda_backward_ws": [c_p] * 9 + [c_i] * 8 + [c_p, c_l, c_p],
    "hipie_msda_backward_workspace": [c_i] * 6,
    "hipie_gemm_batched": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l] + [c_i] * 6 + [c_f, c_p],
    "hipie_gemm_batched_softmax_bias": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l] + [c_i] * 5 + [c_p, c_i, c_p, c_f, c_f, c_p],
    "hipie_gemm_batched_resid": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l] + [c_i] * 5 + [c_f, c_p],
    "hipie_gemm_batched_softmax": [c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l, c_p, c_l, c_l, c_l] + [c_i] * 5 + [c_p, c_i, c_f, c_f, c_p],
    "hipie_softmax_hl8": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_p, c_l, c_f, c_p],
    "hipie_attn_f32": [c_p] * 5 + [c_i] * 5 + [c_l] * 6 + [c_f, c_p],
    "hipie_attn_split": [c_p] * 5 + [c_i] * 5 + [c_l] * 6 + [c_f, c_p],
    "hipie_attn_f32_rows": [c_p] * 5 + [c_i] * 5 + [c_l] * 6 + [c_f, c_p],
    "hipie_attn_split_rows": [c_p] * 5 + [c_i] * 5 + [c_l] * 6 + [c_f, c_p],
    "hipie_conv3x3_split": [c_p, c_l, c_p, c_p, c_p, c_l, c_l] + [c_i] * 6 + [c_p],
    "hipie_ffn_fused": [c_p, c_l, c_p, c_p, c_p, c_p, c_p, c_l, c_i, c_i, c_i, c_p],
    "hipie_topk": [c_p, c_l, c_i, c_i, c_i, c_p, c_p, c_p],
    "hipie_fill_rows": [c_p, c_l, c_p, c_l, c_p, c_l, c_p],
    "hipie_to_hl8": [c_p, c_l, c_p, c_l, c_l, c_i, c_i, c_f, c_p],
    "hipie_to_hl8_t": [c_p, c_l, c_p, c_l, c_l, c_i, c_l, c_f, c_p],
    "hipie_attn_train_forward": [c_p] * 8 + [c_i, c_i, c_p],
    "hipie_to_f16_pair": [c_p, c_l, c_p, c_p, c_l, c_i, c_i, c_p, c_p],
    "hipie_attn_train_backward": [c_p] * 13 + [c_i, c_i, c_p],
    "hipie_selftest": [c_i, c_p, c_p, c_p, c_p],
}

_lib = None


class HipieLibraryError(RuntimeError):
    pass


def load():
    """dlopen the library once and set the prototypes; raises HipieLibraryError when it is absent or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipieLibraryError(
            "libhipie_mi355.so not found at %s -- build it with `make -C hipie_amd/csrc` (hipcc, gfx950). "
            "There is no PyTorch fallback for the hand-written kernels." % LIB_PATH)
    # PyTorch-ROCm ships its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch FIRST so that this library
    # binds to the HIP runtime instance torch has initialised: one runtime per process, shared streams and pointers.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise HipieLibraryError("cannot load %s: %s" % (LIB_PATH, e))
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipieLibraryError("%s does not export %s (stale build?)" % (LIB_PATH, name))
        fn.argtypes = argtypes
        fn.restype = (ctypes.c_char_p if name == "hipie_last_error" else
                      ctypes.c_int64 if name in ("hipie_bi_xattn_workspace", "hipie_mask_einsum_workspace", "hipie_msda_backward_workspace") else ctypes.c_int)
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().hipie_last_error()
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else ""))
