"""DDETRSegmUniDN (inference): orchestration of the two heads and the CondInst dynamic-mask branch (SURVEY rows a17-a19,
a22).  Mirrors hipie/models/ddetrs_dn.py: __init__ (:90-215), coco_inference (:801-978), forward_mask_head_train
(:1006-1069), MaskHeadSmallConv (:1580-1689), post_process_maskdino (:244-262).
The per-instance dynamic conv + aligned_bilinear run on hipie_dynamic_mask.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .maskdino import MaskDINOHead
from .transformer import (MLP, FeatureResizer, NestedTensor, PConv2d, _get_clones, agg_lang_feat, expand_tokens, geo_cached, inverse_sigmoid,
                          nested_tensor_from_images)


_SIDE_STREAMS = {}


def _side_stream(device):
    """one side stream per device for the branch overlap of coco_inference (created on first use)"""
    key = str(device)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class MaskHeadSmallConv(nn.Module):
    """ddetrs_dn.py:1580-1689 with fpn_dims=None, use_raft=False: five 3x3 convs, nearest-upsample adds."""

    def __init__(self, dim, context_dim):
        super().__init__()
        self.lay1 = PConv2d(dim, dim // 4, 3, padding=1)
        self.lay2 = PConv2d(dim // 4, dim // 32, 3, padding=1)
        self.lay3 = PConv2d(context_dim, context_dim, 3, padding=1)
        self.lay4 = PConv2d(context_dim, context_dim, 3, padding=1)
        self.jia_dcn = PConv2d(context_dim, context_dim, 3, padding=1)

    def forward(self, x, fpns=None):
        f = F.relu(self.lay3(x[-1]))
        f = x[-2] + F.interpolate(f, size=x[-2].shape[-2:], mode="nearest")
        f = F.relu(self.lay4(f))
        f = x[-3] + F.interpolate(f, size=x[-3].shape[-2:], mode="nearest")
        f = F.relu(self.jia_dcn(f))
        f = F.relu(self.lay1(f))
        return F.relu(self.lay2(f))


class DDETRSegmUniDN(nn.Module):
    def __init__(self, detr, cfg, precision):
        super().__init__()
        self.detr = detr
        d = cfg.hidden_dim
        self.mask_out_stride = cfg.mask_stride
        self.up_rate = 8 // cfg.mask_stride
        self.num_gen_params = 169                       # (8+2)*8 + 8*8 + 8 weights, 8 + 8 + 1 biases (ddetrs_dn.py:113-131)
        self.controller = MLP(d, d, self.num_gen_params, cfg.ctrl_layers)
        self.mask_head = MaskHeadSmallConv(d, d)
        self.resizer = FeatureResizer(cfg.lang_dim, d)  # DYNAMIC_LABEL_ENC (training-only use; kept for the state_dict)
        self.mask_dino = MaskDINOHead(cfg, detr.backbone.num_channels, precision)
        self.mask_logit_dtype = precision.act      # mask logits leave in the activation dtype (fp32 in the parity policy)
        # 16-bit head policies: the dynamic 10-8-8-1 layers on the matrix pipe (hipie_dynamic_mask16); fp32: the VALU kernel
        # split policy: the same kernel on fp16 pairs (fp32-class)
        self.mask_mlp_dtype = "split" if getattr(precision, "split", False) else (
            precision.head if precision.head in (torch.float16, torch.bfloat16) else None)
        self.feature_keys = ["res3", "res4", "res5"]
        self.mask_dino_cls_embed = _get_clones(self.detr.class_embed[0], cfg.md_dec_layers + 2)
        self.cfg = cfg
        import os
        # the MaskDINO head on a side stream beside the deformable transformer (HIPIE_BRANCH_STREAMS=0: one stream, in order)
        self.overlap_branches = os.environ.get("HIPIE_BRANCH_STREAMS", "1") != "0"

    def post_process_maskdino(self, outputs, language_feat, idx=-1):
        outputs["pred_logits"] = self.mask_dino_cls_embed[idx](outputs["pred_logits"], language_feat)
        return outputs

    def coco_inference(self, samples, gt_targets=None, criterion=None, train=False, language_dict_features=None,
                       task=None, bg_queries_lang=None):
        """samples: object with .image_sizes and iteration over the (unpadded) normalised images (ImageList-like).

        Two streams: the MaskDINO head (pixel decoder + mask decoder) depends on the backbone features only, the deformable
        transformer (input projections, VL fusion, encoder, decoder) likewise, and they meet at the MaskDINO class logits, which need the
        fused language features.  The head runs on a side stream beside the transformer: its many small launches fill the gaps of the
        other branch's (-6 ms per bs-8 step, measured in round 5).  Round 5 had to take this out again because hipie_msda_fused returned
        wrong values beside gemm_kernel<256>; round 6 found the cause -- a gfx950 erratum of packed fp32 VALU instructions with a swapped
        second source when another wave on the SIMD runs MFMAs next to LDS traffic (tools/ubench/pk_f32_hazard.hip)
        -- and removed the instruction form from the library (tests/test_isa_hazards.py, tests/test_gpu_kernels.py::
        test_msda_fused_beside_gemms).  Memory: tensors made on the side stream are consumed on the main stream only behind
        wait_stream, and the next call's side work starts behind side.wait_stream(main), so the caching allocator never hands a block
        to one stream while the other still uses it; kernel workspaces are per stream (ops._Workspace)."""
        assert not train
        image_sizes = samples.image_sizes
        if not isinstance(samples, NestedTensor):
            div = getattr(self.detr.backbone[0].backbone, "size_divisibility", 32)
            samples = nested_tensor_from_images(list(samples), size_divisibility=div, stacked=getattr(samples, "tensor", None))
        features, pos = self.detr.backbone(samples)
        features_maskdino = {k: v.tensors for k, v in zip(self.feature_keys, features)}
        overlap = self.overlap_branches and features[0].tensors.is_cuda
        if overlap:
            main, side = torch.cuda.current_stream(), _side_stream(features[0].tensors.device)
            side.wait_stream(main)                                            # the backbone features (and everything before them)
            with torch.cuda.stream(side):
                outputs_maskdino, _ = self.mask_dino(features_maskdino)
        if task in ("grounding", "sot"):
            lang_feat_pool = agg_lang_feat(language_dict_features["hidden"], language_dict_features["masks"]).unsqueeze(1)
        elif task != "detection":
            raise ValueError("task must be detection or grounding")
        srcs, masks, poses = [], [], []
        for l, feat in enumerate(features):
            src, mask = feat.decompose()
            srcs.append(self.detr.input_proj[l](src))
            masks.append(mask)
            poses.append(pos[l])
        gk = samples.geo_key
        for l in range(len(features), self.detr.num_feature_levels):
            src = self.detr.input_proj[l](features[-1].tensors if l == len(features) else srcs[-1])
            hw = tuple(src.shape[-2:])
            mask = geo_cached(gk, ("level_mask0", hw),
                              lambda: F.interpolate(masks[0][None].float(), size=hw).to(torch.bool)[0])   # level-0 mask (:841)
            srcs.append(src)
            masks.append(mask)
            poses.append(geo_cached(gk, ("pos0", hw, src.dtype), lambda: self.detr.backbone[1](mask).to(src.dtype)))
        hs, memory, init_reference, inter_references, _, _, language_dict_features, spatial_shapes = \
            self.detr.transformer(srcs, masks, poses, language_dict_features, task=task, geo_key=gk)

        lang = lang_feat_pool if task in ("grounding", "sot") else language_dict_features["hidden"]
        if overlap:
            main.wait_stream(side)
        else:
            outputs_maskdino, _ = self.mask_dino(features_maskdino)
        outputs_maskdino = self.post_process_maskdino(outputs_maskdino, lang)
        full_len = None if task in ("grounding", "sot") else language_dict_features.get("full_len")     # PAD_MAX columns dropped by the text encoder
        outputs_maskdino["pred_logits"] = expand_tokens(outputs_maskdino["pred_logits"], full_len)

        outputs = {}
        lvl = hs.shape[0] - 1
        reference = inverse_sigmoid(init_reference if lvl == 0 else inter_references[lvl - 1])
        outputs["pred_logits"] = expand_tokens(self.detr.class_embed[lvl](hs[lvl], lang), full_len)
        outputs["pred_boxes"] = (self.detr.bbox_embed[lvl](hs[lvl]) + reference).sigmoid()
        outputs["pred_boxious"] = self.detr.iou_head[lvl](hs[lvl])
        outputs["reference_points"] = inter_references[-2, :, :, :2]
        params = self.controller(hs[lvl])                                     # (bs, nq, 169)
        bs, nq, _ = params.shape
        key = (tuple(image_sizes), str(params.device))
        if getattr(self, "_scale_key", None) != key:                          # cached: no host->device copy per call
            self._scale = torch.tensor([[float(w), float(h)] for (h, w) in image_sizes], device=params.device)   # (bs,2) = (w,h)
            self._scale_key = key
        scale = self._scale
        ref_px = (outputs["reference_points"] * scale[:, None, :]).reshape(bs * nq, 2)
        outputs["pred_masks"] = self.forward_mask_head(memory, spatial_shapes, ref_px, params.reshape(bs * nq, -1), bs, nq)
        outputs["pred_masks_maskdino"] = outputs_maskdino["pred_masks"]
        outputs["pred_logits_maskdino"] = outputs_maskdino["pred_logits"]
        outputs["pred_boxes_maskdino"] = outputs_maskdino["pred_boxes"]
        return outputs, None

    def forward_mask_head(self, feats, spatial_shapes, reference_points, mask_head_params, bs, nq):
        """forward_mask_head_train + dynamic_mask_with_coords (ddetrs_dn.py:1006-1069, 1411-1502)."""
        c = feats.shape[-1]
        enc, st = [], 0
        for (h, w) in spatial_shapes[:self.detr.num_feature_levels - 1]:
            enc.append(feats[:, st:st + h * w, :].reshape(bs, h, w, c).permute(0, 3, 1, 2))
            st += h * w
        mask_feats = self.mask_head(enc, fpns=None)                           # (bs, 8, H/8, W/8)
        logits = ops.dynamic_mask(mask_feats.float().contiguous(), reference_points.float().contiguous(),
                                  mask_head_params.float().contiguous(), nq, stride=8, up=self.up_rate,
                                  out_dtype=self.mask_logit_dtype, mlp_dtype=self.mask_mlp_dtype)
        return logits.view(bs, nq, 1, logits.shape[-2], logits.shape[-1])
