"""autograd Functions over the hand-written kernels that have a backward.  Forward and backward both run on
libhipie_mi355.so; there is no CPU implementation (the ops raise on host tensors)."""
import torch

from .. import ops


class MaskEinsumFunction(torch.autograd.Function):
    """einsum("bqc,bchw->bqhw") [+ row_bias[b, q]]: MaskDINO's mask logits (maskdino_decoder.py forward_prediction_heads) and the
    FoldedMaskFeatures form of the inference path.  forward = hipie_mask_einsum (three-product split), backward = two
    hipie_gemm_batched products (ops.mask_einsum_backward) + a row sum for the bias."""

    @staticmethod
    def forward(ctx, mask_embed, mask_features, row_bias=None):
        ctx.save_for_backward(mask_embed, mask_features)
        ctx.has_bias = row_bias is not None
        return ops.mask_einsum(mask_embed.float().contiguous(), mask_features.float().contiguous(), precision=1, row_bias=row_bias)

    @staticmethod
    def backward(ctx, grad_out):
        e, f = ctx.saved_tensors
        ge, gf = ops.mask_einsum_backward(e, f, grad_out.contiguous())
        gb = grad_out.sum((2, 3)) if ctx.has_bias else None
        return ge.to(e.dtype), gf.to(f.dtype), gb


def mask_einsum(mask_embed, mask_features, row_bias=None):
    return MaskEinsumFunction.apply(mask_embed, mask_features, row_bias)


class DynamicMaskFunction(torch.autograd.Function):
    """the CondInst dynamic mask head (DDETRSegmUniDN.dynamic_mask_with_coords, models/ddetrs_dn.py:1411-1502) with gradients for the
    mask features, the reference points and the controller parameters: forward = hipie_dynamic_mask (fp32 kernel), backward =
    hipie_dynamic_mask_backward.  mask_feats (B,8,H,W), ref_points (B*Q,2) pixels, params (B*Q,169) -> (B*Q, up*H, up*W)."""

    @staticmethod
    def forward(ctx, mask_feats, ref_points, params, num_queries, stride=8, up=2):
        mask_feats, ref_points, params = mask_feats.float().contiguous(), ref_points.float().contiguous(), params.float().contiguous()
        ctx.save_for_backward(mask_feats, ref_points, params)
        ctx.geom = (int(num_queries), int(stride), int(up))
        return ops.dynamic_mask(mask_feats, ref_points, params, num_queries, stride=stride, up=up)

    @staticmethod
    def backward(ctx, grad_out):
        feats, refs, params = ctx.saved_tensors
        q, stride, up = ctx.geom
        gf, gr, gp = ops.dynamic_mask_backward(feats, refs, params, grad_out.float().contiguous(), q, stride=stride, up=up)
        return gf, gr, gp, None, None, None


def dynamic_mask(mask_feats, ref_points, params, num_queries, stride=8, up=2):
    return DynamicMaskFunction.apply(mask_feats, ref_points, params, num_queries, stride, up)
