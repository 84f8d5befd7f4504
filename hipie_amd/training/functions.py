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


def _weight_grad(g2, x2):
    """dW = dy^T . x  (N x K, the contraction over the M token rows) on the split GEMM.  N x K is 25-100 tiles of 256 x 256 for the ViT-H
    linears -- a fraction of the 256 CUs -- so the M rows are cut into nk chunks, one problem each in ONE hipie_gemm_batched launch, and
    the partial products are summed (the 64 x 128-tile kernel the single problem fell to ran at 160 TFLOP/s: 0.51 ms per linear)."""
    M, N = g2.shape
    K = x2.shape[1]
    Mp = -(-M // 32) * 32
    tiles = -(-N // 256) * -(-K // 256)
    nk = 1
    while nk < 16 and tiles * nk < 256 and (Mp // 32) % (2 * nk) == 0:
        nk *= 2
    gt, xt = ops.to_hl8_t(g2, 32), ops.to_hl8_t(x2, 32)                                           # (N | K, 2 Mp) fp16 pairs, one pass each
    if nk == 1:
        return ops.gemm(gt, xt, None, split=True, out_fmt=ops.F32, tag="train_dw")
    return ops.gemm_split_k(gt, xt, nk)


class SplitLinearFunction(torch.autograd.Function):
    """F.linear(x, weight, bias) with forward AND backward on hipie_gemm's split-fp16 operands (three MFMA products, fp32 accumulation:
    fp32-class results at ~2.7x the rate of the fp32 matrix pipe): the linears of the training step that carry its flops (ViT qkv / proj /
    fc1 / fc2, the encoder FFNs).
        y  = x . W^T + b            hipie_gemm(A = x fp32 rows, W as HL8)
        dx = dy . W                 hipie_gemm(A = dy fp32 rows, W^T as HL8)
        dW = dy^T . x               split operands dy^T and x^T, the contraction over the M rows (padded to 32) cut into chunks that run as
                                    one hipie_gemm_batched launch (_weight_grad)
        db = sum_m dy
    The two transposed operands cost one pass each; the HL8 copies of W and W^T are cached per parameter version on `owner`."""

    @staticmethod
    def forward(ctx, x, weight, bias, owner, key):
        w_hl8, _, _ = ops.split_weight(owner, key, [weight], lambda: weight)
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        ctx.save_for_backward(x2, weight)
        ctx.owner, ctx.key, ctx.lead, ctx.has_bias = owner, key, x.shape[:-1], bias is not None
        y = ops.gemm(x2, w_hl8, None if bias is None else bias.detach().float().contiguous(), split=True, out_fmt=ops.F32, tag="train_fwd")
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, gy):
        x2, weight = ctx.saved_tensors
        N, K = weight.shape
        g2 = gy.reshape(-1, N).float()
        if g2.stride(-1) != 1:
            g2 = g2.contiguous()
        M = g2.shape[0]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wt_hl8, _, _ = ops.split_weight(ctx.owner, ctx.key + ".T", [weight], lambda: weight.t().contiguous())
            gx = ops.gemm(g2, wt_hl8, None, split=True, out_fmt=ops.F32, tag="train_dx").view(*ctx.lead, K)
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(g2, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb, None, None


def split_linear(x, weight, bias, owner, key):
    """F.linear on the split GEMM with a backward (SplitLinearFunction); shapes it does not cover fall through to the library"""
    K, N = weight.shape[1], weight.shape[0]
    if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and K % 32 == 0 and N % 32 == 0 and x.numel() // K >= 256:
        return SplitLinearFunction.apply(x, weight, bias, owner, key)
    return torch.nn.functional.linear(x, weight, bias)


class FusedAttentionFunction(torch.autograd.Function):
    """softmax(q' k'^T) v with the decomposed rel-pos bias folded into q' / k' (net.vit_attention), forward and backward on
    hipie_attn_train_forward / _backward (csrc/attn_train.hip): no (heads, N, N) tensor in HBM.  q' (BH, N, <= 224), k' (BH, N, <= 224) whose
    columns from 80 on are CONSTANT (the key-axis indicators: they get a zero gradient), v (BH, N, 80), N a multiple of 128.
    Saved for the backward: the fp16 pairs of q' and k' (the bytes of the fp32 operands), v, the output and the log-sum-exp row."""

    @staticmethod
    def forward(ctx, qa, ka, v):
        qp, kp = ops.f16_pair(qa, 224), ops.f16_pair(ka, 224)
        out, lse = ops.attn_train_forward(qp, kp, ops.f16_pair(v))
        ctx.save_for_backward(qp[0], qp[1], kp[0], kp[1], v, out, lse)
        ctx.cols = (qa.shape[-1], ka.shape[-1])
        return out

    @staticmethod
    def backward(ctx, go):
        qh, ql, kh, kl, v, out, lse = ctx.saved_tensors
        go = go.float().contiguous()
        # dO enters the kernels as fp16 pairs: scaled by a power of two so that its largest entry sits in [8, 16) (a device scalar: no host
        # wait) -- high enough for the pairs of dO and dS = P (dP - delta) to be normal fp16 numbers, low enough for |dP| <= 16 * 80 * max|v|
        scale = torch.exp2(torch.floor(torch.log2(16.0 / go.abs().amax().clamp_min(1e-30))))
        delta = (go * out).sum(-1) * scale
        dq, dk, dv = ops.attn_train_backward((qh, ql), (kh, kl), ops.f16_pair(v, 96), ops.f16_pair(go, 96, scale), lse, delta)
        inv = 1.0 / scale
        cq, ck = ctx.cols
        return dq[..., :cq] * inv, torch.nn.functional.pad(dk * inv, (0, ck - 80)), dv * inv


def fused_attention_ok(qa, ka, v):
    """operands FusedAttentionFunction takes as they are (the global blocks of the ViT at grids whose token count is a multiple of 128)"""
    return (qa.is_cuda and qa.dtype == torch.float32 and v.shape[-1] == 80 and qa.shape[-1] <= 224 and ka.shape[-1] == qa.shape[-1]
            and qa.shape[1] % 128 == 0)


def fused_attention(qa, ka, v):
    """softmax(q' k'^T) v on the fused kernels, or None when the operands are not covered.  Token counts that are not a multiple of 128 (a
    50 x 76 grid: 3800) are padded: the padding KEYS get a bias of -30000 through one more operand column (q' column 1, k' column -30000
    on the padding rows) so they receive probability 0; the padding QUERY rows are zeros and their outputs are dropped (no gradient
    reaches them)."""
    BH, N, C = qa.shape
    if N % 128 == 0:
        return FusedAttentionFunction.apply(qa, ka, v) if fused_attention_ok(qa, ka, v) else None
    if N < 1024:
        # the 196-token windows: correct on this path (tests) but SLOWER than the materialised formulation -- the kernels always run 224
        # operand columns (a window has 108) and 256 rows: 752 against 720 ms per training step
        return None
    Np = -(-N // 128) * 128
    pad = torch.nn.functional.pad
    if not fused_attention_ok(pad(qa[:, :0], (0, 1, 0, Np)), pad(ka[:, :0], (0, 1, 0, Np)), pad(v[:, :0], (0, 0, 0, Np))):
        return None
    bias = ka.new_zeros(BH, Np, 1)
    bias[:, N:] = -30000.0
    qa_p = pad(torch.cat((qa, qa.new_ones(BH, N, 1)), -1), (0, 0, 0, Np - N))
    ka_p = torch.cat((pad(ka, (0, 0, 0, Np - N)), bias), -1)
    return FusedAttentionFunction.apply(qa_p, ka_p, pad(v, (0, 0, 0, Np - N)))[:, :N]
