"""MaskDINO "stuff/panoptic" branch: pixel decoder (6-layer MSDeformAttn encoder + one FPN
level + mask_features) and the two-stage DINO decoder whose mask logits are the query x pixel contraction.

Mirrors hipie/models/maskdino/pixel_decoder/maskdino_encoder.py (MaskDINOEncoder, feature_order low2high),
transformer_decoder/maskdino_decoder.py (MaskDINODecoder, eval path), transformer_decoder/dino_decoder.py
(TransformerDecoder) and meta_arch/maskdino_head.py (MaskDINOHead) with the reference's parameter names.
The branch is called with mask=None (ddetrs_dn.py:885): all-False padding masks, valid_ratio 1.
"""
import collections
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .transformer import (MLP, _select_topk, DeformableTransformerDecoderLayer, DeformableTransformerEncoderLayer, FeatureResizer, geo_cached,
                          PConv2d, PGroupNorm, PLayerNorm, PLinear, PositionEmbeddingSine, _get_clones, encoder_reference_points,
                          gen_encoder_output_proposals, level_tensors,
                          batched_decoder_values, decoder_box_refine, decoder_split_values, selected_proposal_boxes, decoder_fast_path, decoder_query_pos)


class NormConv2d(PConv2d):
    """detectron2.layers.Conv2d: conv (no bias) + norm (+ activation); parameter names <name>.weight, <name>.norm.*"""

    def __init__(self, cin, cout, k, padding=0, relu=False):
        super().__init__(cin, cout, kernel_size=k, padding=padding, bias=False)
        self.norm = PGroupNorm(32, cout)
        self.relu = relu

    def forward(self, x):
        return self.norm(super().forward(x), relu=self.relu)          # the ReLU rides in the GroupNorm pass


class FoldedMaskFeatures(object):
    """the mask_features head without its last 1x1 convolution (weight (256,256), bias (256)): `pre` (B,256,H/4,W/4) NCHW, fp32 or
    the 16-bit activation dtype."""

    def __init__(self, pre, weight, bias, owner=None):
        self.pre, self.weight, self.bias, self.owner = pre, weight, bias, owner

    def fold(self, e32):
        """(emb . W, emb . b) for emb (B, Q, C_out): the query side of the folded convolution.  On the device with an owning module (the
        convolution, which carries the cache of the split weight): ONE hipie_gemm whose extra output row is the bias vector."""
        w, b = self.weight, self.bias
        if self.owner is not None and e32.is_cuda and w.dtype == torch.float32 and ops.split_ok(w.shape[0]):
            cin = w.shape[1]
            y = ops.split_linear(e32.contiguous(), self.owner, "fold_aug", self.owner.weight, None,
                                 weight_fn=lambda: torch.cat([w.float().t(), b.float()[None]], 0), params=[self.owner.weight, self.owner.bias])
            return y[..., :cin].contiguous(), y[..., cin].contiguous()
        return (e32 @ w.float()).contiguous(), e32 @ b.float()


class MSDeformAttnTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = _get_clones(encoder_layer, num_layers)


class MSDeformAttnTransformerEncoderOnly(nn.Module):
    """maskdino_encoder.py:42-115."""

    def __init__(self, d_model, nhead, num_layers, dim_feedforward, num_levels, value_dtype):
        super().__init__()
        layer = DeformableTransformerEncoderLayer(d_model, dim_feedforward, num_levels, nhead, 4, value_dtype)
        self.encoder = MSDeformAttnTransformerEncoder(layer, num_layers)
        self.level_embed = nn.Parameter(torch.randn(num_levels, d_model))

    def forward(self, srcs, pos_embeds):
        shapes_list = [tuple(int(v) for v in s.shape[-2:]) for s in srcs]
        src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
        B = src.shape[0]
        # all-valid masks: the geometry is (batch, level shapes) alone
        gk = ("md_enc", B, tuple(shapes_list), str(src.device), src.dtype)
        if not hasattr(self, "_own_cache"):
            self._own_cache = collections.OrderedDict()
        pos = geo_cached(gk, "pos_flat", lambda: torch.cat(
            [p.flatten(2).transpose(1, 2) + self.level_embed[i].view(1, 1, -1) for i, p in enumerate(pos_embeds)], 1).to(src.dtype),
            store=self._own_cache)
        spatial_shapes, level_start_index = level_tensors(shapes_list, src.device)
        refs = geo_cached(gk, "enc_refs", lambda: encoder_reference_points(
            shapes_list, torch.ones(B, len(srcs), 2, device=src.device), src.device))
        q, n = None, len(self.encoder.layers)
        for i, layer in enumerate(self.encoder.layers):
            if i + 1 < n:                                 # the last LayerNorm pass of a layer also emits the next layer's src + pos
                src, q = layer(src, pos, refs, spatial_shapes, level_start_index, None, query=q, want_query=True)
            else:
                src = layer(src, pos, refs, spatial_shapes, level_start_index, None, query=q)
        return src, shapes_list


class MaskDINOEncoder(nn.Module):
    """maskdino_encoder.py:190-434 for input features {res3,res4,res5}, 4 total levels, low2high, GN, num_fpn_levels 1."""

    def __init__(self, cfg, in_channels, precision):
        super().__init__()
        cd = cfg.md_conv_dim
        # input_proj[0..2] act on reversed transformer_in_features = [res3, res4, res5]; [3] = stride-2 conv on res5
        self.input_proj = nn.ModuleList(
            [nn.Sequential(PConv2d(c, cd, kernel_size=1), PGroupNorm(32, cd)) for c in in_channels] +
            [nn.Sequential(PConv2d(max(in_channels), cd, kernel_size=3, stride=2, padding=1), PGroupNorm(32, cd))])
        self.transformer = MSDeformAttnTransformerEncoderOnly(cd, 8, cfg.md_enc_layers, cfg.md_enc_dim_feedforward, 4, precision.value)
        self.pe_layer = PositionEmbeddingSine(cd // 2, offset=0.0)
        self.mask_features = nn.Sequential(nn.ConvTranspose2d(cd, cd, 2, stride=2), PGroupNorm(32, cd), nn.ReLU(),
                                           PConv2d(cd, cfg.md_mask_dim, kernel_size=1))
        self.adapter_1 = NormConv2d(in_channels[0], cd, 1)
        self.layer_1 = NormConv2d(cd, cd, 3, padding=1, relu=True)
        self.precision = precision

    def _mask_features_front(self, z, out_dtype=None):
        """ConvTranspose2d -> GroupNorm -> ReLU of the mask_features head (maskdino_encoder.py:289-292): the transposed conv runs
        without its bias, which enters the GroupNorm pass as a per-channel pre-bias together with the ReLU (one read + one write
        of the (B,256,H/4,W/4) map instead of the bias, norm and ReLU passes)."""
        ct, gn = self.mask_features[0], self.mask_features[1]
        C = ct.weight.shape[0]              # any layout in (channels-last from the 3x3 conv + GroupNorm in front), NCHW out: the map leaves
        if getattr(self.precision, "split", False) and z.is_cuda and ct.weight.dtype == torch.float32 and ops.split_ok(C):   # pixel-fastest
            # split policy: ConvTranspose2d(k = 2, s = 2) is ONE linear per input pixel (C -> 4 C: tap-major columns) + a pixel shuffle;
            # the linear runs on hipie_gemm's split operands (the library's fp32 transposed convolution took 1.26 ms here, this 0.5)
            B, _, H, W = z.shape
            wt = ct.weight                                                   # (C_in, C_out, 2, 2)
            rows = z.permute(0, 2, 3, 1).reshape(B * H * W, C)
            # (ops.convt2x2_split: one linear per tap, written through a row map into the up-sampled channels-last map -- no shuffle pass)
            y = ops.convt2x2_split(rows, ct, "convt", wt, None, B, H, W).permute(0, 3, 1, 2)
        else:
            y = F.conv_transpose2d(z.contiguous().to(ct.weight.dtype), ct.weight, None, ct.stride, ct.padding, ct.output_padding, ct.groups, ct.dilation)
        if out_dtype is not None:
            y = y.to(out_dtype)
        return gn(y, relu=True, prebias=ct.bias.float(), out_nchw=True)   # pixel-fastest: the mask contraction's operand layout (callers `.contiguous()` it)

    def forward_features(self, features, masks=None):
        f3, f4, f5 = features["res3"], features["res4"], features["res5"]      # the projections cast / lay out their input
        extra = self.input_proj[3](f5)
        srcs = [self.input_proj[i](f) for i, f in enumerate((f3, f4, f5))] + [extra]
        pos = [geo_cached(("md_pe", tuple(s.shape[0:1]) + tuple(s.shape[2:]), str(s.device)), "pos", lambda s=s: self.pe_layer(
            torch.zeros(s.shape[0], s.shape[2], s.shape[3], dtype=torch.bool, device=s.device))) for s in srcs]
        y, shapes = self.transformer(srcs, pos)
        B = y.shape[0]
        out, st = [], 0
        y = y.contiguous()
        for (H, W) in shapes:
            # (B,C,H,W) VIEWS of the token-major encoder memory (channels-last strides): the decoder flattens them straight back to
            # tokens and the FPN sum below reads them in place -- no (B,256,H,W) transposes through HBM in either direction
            out.append(y[:, st:st + H * W].transpose(1, 2).unflatten(2, (H, W)))
            st += H * W
        cur = self.adapter_1(f3)
        top = out[0]
        if tuple(top.shape[-2:]) != tuple(cur.shape[-2:]):      # same size (always, for /32-padded inputs): bilinear resampling with
            top = F.interpolate(top, size=cur.shape[-2:], mode="bilinear", align_corners=False)   # align_corners=False is the identity
        z = cur + top
        z = self.layer_1(z)
        if self.precision.einsum >= 3:
            # 16-bit policies: stop in front of the head's last 1x1 convolution.  mask logits = emb . (W x + b) = (emb . W) . x +
            # emb . b, so that convolution folds into the (tiny) query side of the contraction (forward_prediction_heads): no
            # (B,256,H/4,W/4) GEMM, no fp32 round trips; x leaves NCHW (pixel fastest, the contraction's operand layout) in `act`
            x = self._mask_features_front(z)
            conv = self.mask_features[3]
            mf = FoldedMaskFeatures(x.to(self.precision.act).contiguous(), conv.weight.reshape(conv.weight.shape[0], -1), conv.bias)
        elif self.precision.einsum in (1, 2) and z.is_cuda:
            # fp32 features on the split contraction: the same fold (hipie_mask_einsum_bias adds emb . b per query row), so the
            # library's 1x1 convolution over the (B,256,H/4,W/4) map (0.76 ms at the headline shape) and its second copy of the
            # map disappear; x stays fp32, NCHW
            x = self._mask_features_front(z, torch.float32)
            conv = self.mask_features[3]
            mf = FoldedMaskFeatures(x.float().contiguous(), conv.weight.reshape(conv.weight.shape[0], -1), conv.bias,
                                    owner=conv if getattr(self.precision, "split", False) else None)
        else:
            mf = self.mask_features[3](self._mask_features_front(z, torch.float32))
            mf = mf.float().contiguous()  # NCHW fp32, pixel fastest: hipie_mask_einsum's operand layout, produced once for both calls
        return mf, out[0], out          # mask_features (B,256,H/4,W/4), s8 level, [s8,s16,s32,s64]


class TransformerDecoder(nn.Module):
    """dino_decoder.py:19-168 (deformable, query_dim 4, shared bbox head, norm on every intermediate)."""

    def __init__(self, decoder_layer, num_layers, norm, d_model):
        super().__init__()
        self.layers = _get_clones(decoder_layer, num_layers)
        self.norm = norm
        self.ref_point_head = MLP(2 * d_model, d_model, d_model, 2)
        self.bbox_embed = None


class MaskDINODecoder(nn.Module):
    """maskdino_decoder.py:36-529, eval path (two_stage, initial_pred, initialize_box_type 'no', learn_tgt False)."""

    def __init__(self, cfg, precision):
        super().__init__()
        d = cfg.hidden_dim
        self.num_queries, self.num_layers, self.hidden_dim = cfg.md_num_queries, cfg.md_dec_layers, d
        self.enc_output = PLinear(d, d)
        self.enc_output_norm = PLayerNorm(d)
        self.class_embed = PLinear(d, d)                     # num_classes = hidden_dim (ddetrs_dn.py:183-185)
        self.resizer = FeatureResizer(768, d)                # dynamic_label_enc (training only; kept for the state_dict)
        self.mask_embed = MLP(d, d, cfg.md_mask_dim, 3)
        self.decoder_norm = PLayerNorm(d)
        layer = DeformableTransformerDecoderLayer(d, cfg.md_dim_feedforward, 4, 8, 4, precision.value, precision.attn)
        self.decoder = TransformerDecoder(layer, self.num_layers, self.decoder_norm, d)
        self._bbox_embed = MLP(d, d, 4, 3)
        self.bbox_embed = nn.ModuleList([self._bbox_embed for _ in range(self.num_layers)])
        self.decoder.bbox_embed = self.bbox_embed
        self.precision = precision
        self.pinned_topk = None
        self.last_topk = None
        # the reference's eval path also evaluates the mask contraction for the two-stage proposals (interm_outputs, consumed
        # only by the training losses); kept on so the path does the same work -- set False to drop that 0.46 ms launch
        self.initial_pred_masks = True

    def forward_prediction_heads(self, output, mask_features, pred_mask=True):
        """maskdino_decoder.py:520-529 -- the mask-logit contraction runs on hipie_mask_einsum."""
        dec = self.decoder_norm(output)
        cls = self.class_embed(dec)
        masks = None
        if pred_mask:
            emb = self.mask_embed(dec)
            if isinstance(mask_features, FoldedMaskFeatures):      # 16-bit features: 3 = single product, 4 = embedding split hi + lo; fp32: 1 | 2
                e32 = emb.float()
                pre = mask_features.pre
                ew, eb = mask_features.fold(e32)
                if pre.dtype == torch.float32:                      # fp32 features: the bf16-split contraction with a row bias
                    masks = ops.mask_einsum(ew, pre, precision=self.precision.einsum, out_dtype=self.precision.act, row_bias=eb)
                elif e32.shape[1] <= 320 and (pre.shape[-1] * pre.shape[-2]) % 8 == 0:
                    masks = ops.mask_einsum16(ew, pre, split=self.precision.einsum == 4, row_bias=eb)
                else:       # more queries than the 16-bit kernel's tile (or an odd pixel count): the fp32-feature kernel, bias added after
                    masks = ops.mask_einsum(ew, pre.float().contiguous(), precision=1, out_dtype=self.precision.act)
                    masks = masks + eb.to(masks.dtype)[..., None, None]
            else:
                masks = ops.mask_einsum(emb.float().contiguous(), mask_features, precision=self.precision.einsum,
                                        out_dtype=self.precision.act)
        return cls, masks

    def forward(self, x, mask_features):
        nl = len(x)
        xs = [x[nl - 1 - i] for i in range(nl)]                         # memory order [s64,s32,s16,s8] (:385-397)
        shapes_list = [tuple(int(v) for v in t.shape[-2:]) for t in xs]
        src = torch.cat([t.flatten(2).transpose(1, 2) for t in xs], 1)
        B = src.shape[0]
        gk = ("md_dec", B, tuple(shapes_list), str(src.device))
        mask = geo_cached(gk, "mask_flat", lambda: torch.zeros(B, src.shape[1], dtype=torch.bool, device=src.device))
        spatial_shapes, level_start_index = level_tensors(shapes_list, src.device)
        vr2 = geo_cached(gk, "vr2", lambda: torch.ones(B, 1, nl, 4, device=src.device))
        om, prop = gen_encoder_output_proposals(src, mask, shapes_list, gk)
        om = self.enc_output_norm(self.enc_output(om))
        cls_un = self.class_embed(om)
        if self.pinned_topk is not None:
            topk = self.pinned_topk.to(src.device)
        else:
            topk = _select_topk(cls_un.max(-1)[0], self.num_queries)
        self.last_topk = topk
        # the box head on the selected proposals only (maskdino_decoder.py:396-405 heads every token, then gathers: the same rows)
        ref_un = selected_proposal_boxes(self._bbox_embed, om, prop, topk)
        tgt = torch.gather(om, 1, topk.unsqueeze(-1).repeat(1, 1, self.hidden_dim))
        # einsum #1 (:428): interm_outputs
        interm_cls, interm_mask = self.forward_prediction_heads(tgt, mask_features, pred_mask=self.initial_pred_masks)
        ref = ref_un.sigmoid()
        sdt = torch.float32                                  # query stream dtype: fp32 in every policy
        wdt = self.decoder.ref_point_head.layers[0].weight.dtype
        refs, out, hs = [ref], tgt.to(sdt), []
        layers = self.decoder.layers
        if decoder_fast_path(layers, sdt) and wdt == layers[0].linear1.weight.dtype:
            values = batched_decoder_values(self.decoder, layers, src, None)
            t32 = out.contiguous()
            t16 = t32.to(wdt)
            for lid, layer in enumerate(layers):
                ref_in = ref[:, :, None] * vr2
                qp16 = decoder_query_pos(self.decoder.ref_point_head, ref_in[:, :, 0, :], wdt)
                t32, t16 = layer.forward16(t32, t16, qp16, ref_in, values[lid], spatial_shapes, level_start_index)
                ref = decoder_box_refine(self.decoder.bbox_embed[lid], t32, ref)
                refs.append(ref)
            hs.append(self.decoder.norm(t32).float())
            layers = []
        values = batched_decoder_values(self.decoder, layers, src, None) if layers and decoder_split_values(self.decoder, src) else None
        for lid, layer in enumerate(layers):
            ref_in = ref[:, :, None] * vr2
            query_pos = self.decoder.ref_point_head(ops.sine_embed(ref_in[:, :, 0, :], out_dtype=wdt))
            out = layer(out, query_pos, ref_in, src, spatial_shapes, level_start_index, None, value=None if values is None else values[lid])
            new_ref = ops.box_refine(self.decoder.bbox_embed[lid](out), ref)
            ref = new_ref
            refs.append(new_ref)
            if lid == len(self.decoder.layers) - 1:        # eval: heads on the last layer only (maskdino_decoder.py:485)
                hs.append(self.decoder.norm(out).float())
        cls, masks = self.forward_prediction_heads(hs[-1], mask_features)                 # einsum #2 (:485)
        boxes = ops.box_refine(self.bbox_embed[-1](hs[-1]), refs[-2])                     # pred_box (:357-375)
        return {"pred_logits": cls, "pred_masks": masks, "pred_boxes": boxes,
                "interm_outputs": {"pred_logits": interm_cls, "pred_masks": interm_mask, "pred_boxes": ref_un.sigmoid()}}


class MaskDINOHead(nn.Module):
    """meta_arch/maskdino_head.py:21-82."""

    def __init__(self, cfg, in_channels, precision):
        super().__init__()
        self.pixel_decoder = MaskDINOEncoder(cfg, in_channels, precision)
        self.predictor = MaskDINODecoder(cfg, precision)

    def forward(self, features, mask=None):
        mask_features, _, multi_scale = self.pixel_decoder.forward_features(features, mask)
        return self.predictor(multi_scale, mask_features), None
