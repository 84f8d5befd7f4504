"""Drop-in for the reference's pybind extension ``MultiScaleDeformableAttention`` (ops/src/vision.cpp:13-16).

``install()`` puts a module object of that name into ``sys.modules`` exposing ``ms_deform_attn_forward`` and
``ms_deform_attn_backward`` with the reference's positional signatures, so the reference's own ``MSDeformAttnFunction``
(ops/functions/ms_deform_attn_func.py:21-41, both copies) runs forward AND backward on libhipie_mi355 unchanged.
``MSDeformAttnFunction`` below is the same autograd wrapper for callers that do not import the reference's.
"""
import sys
import types

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step):
    return ops.ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step)


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step):
    """-> (grad_value, grad_sampling_loc, grad_attn_weight), the tuple the reference's binding returns (ms_deform_attn.h:42-62)."""
    return ops.ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step)


class MSDeformAttnFunction(Function):
    """ops/functions/ms_deform_attn_func.py:21-41 on hipie_msda_forward / hipie_msda_backward."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        out = ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, level_start, loc, attn = ctx.saved_tensors
        gv, gl, ga = ms_deform_attn_backward(value, shapes, level_start, loc, attn, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


def install(name="MultiScaleDeformableAttention"):
    mod = types.ModuleType(name)
    mod.ms_deform_attn_forward = ms_deform_attn_forward
    mod.ms_deform_attn_backward = ms_deform_attn_backward
    mod.__doc__ = __doc__
    sys.modules[name] = mod
    return mod
