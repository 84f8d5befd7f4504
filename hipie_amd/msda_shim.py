"""Drop-in for the reference's pybind extension ``MultiScaleDeformableAttention`` (ops/src/vision.cpp:13-16).

``install()`` puts a module object of that name into ``sys.modules`` exposing ``ms_deform_attn_forward`` with the
reference's positional signature, so the reference's own ``MSDeformAttnFunction.forward``
(ops/functions/ms_deform_attn_func.py:21-30, both copies) runs on libhipie_mi355 unchanged.  ``ms_deform_attn_backward``
raises: training is out of scope.
"""
import sys
import types

from . import ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    return ops.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def ms_deform_attn_backward(*args, **kwargs):
    raise NotImplementedError("hipie_amd implements the inference path only (ms_deform_attn_backward is training)")


def install(name="MultiScaleDeformableAttention"):
    mod = types.ModuleType(name)
    mod.ms_deform_attn_forward = ms_deform_attn_forward
    mod.ms_deform_attn_backward = ms_deform_attn_backward
    mod.__doc__ = __doc__
    sys.modules[name] = mod
    return mod
