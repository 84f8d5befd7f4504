"""HIPIE_IMG meta-architecture (inference): the drop-in boundary b1 of SURVEY 8b.

Mirrors hipie/hipie_img.py: __init__ (:51-262, the parts the eval path touches), forward (:263-420, eval branch),
preprocess_image (:880-898), forward_text (:900-922).  ``forward_raw`` returns the a22 parity surface
(DDETRSegmUniDN.coco_inference's output dict); ``forward`` adds the per-image post-processing.

Construction needs no detectron2: pass a HipieConfig (hipie_amd/config.py).  With detectron2 installed,
hipie_amd.d2_registry registers this class as META_ARCH "HIPIE_IMG" with the reference's ``__init__(cfg)`` signature.
"""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .config import HipieConfig, Precision
from .modeling.ddetrs_dn import DDETRSegmUniDN
from .modeling.text import BertEncoder
from .modeling.transformer import (DeformableDETRDINO, DeformableTransformerVLDINO, Joiner, MaskedBackbone,
                                   PositionEmbeddingSine, cast_head, set_split)
from .modeling.vit import D2ViT


class ImageList(object):
    """the slice of detectron2.structures.ImageList the path uses: padded batch + per-image sizes."""

    def __init__(self, tensor, image_sizes):
        self.tensor, self.image_sizes = tensor, image_sizes

    def __len__(self):
        return len(self.image_sizes)

    def __iter__(self):
        for t, (h, w) in zip(self.tensor, self.image_sizes):
            yield t[:, :h, :w]

    @staticmethod
    def from_tensors(tensors):
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        out = tensors[0].new_zeros(len(tensors), tensors[0].shape[0], H, W)
        for i, t in enumerate(tensors):
            out[i, :, :t.shape[-2], :t.shape[-1]].copy_(t)
        return ImageList(out, sizes)


def build_backbone(cfg, precision):
    if cfg.backbone == "vit":
        return D2ViT(cfg, precision)
    if cfg.backbone == "r50":
        from .modeling.resnet import ResNet50
        return ResNet50(precision)
    raise ValueError("unknown backbone %r" % cfg.backbone)


class HIPIE_IMG(nn.Module):
    def __init__(self, cfg: HipieConfig, precision: Precision = None, device="cuda"):
        super().__init__()
        _lib.load()       # fail loudly at construction when the HIP library is missing -- there is no fallback path
        self.cfg = cfg
        self.precision = precision or Precision()
        self.device = torch.device(device)
        self.demo_only = False
        self.mask_stride = cfg.mask_stride
        bb = build_backbone(cfg, self.precision)
        d2_backbone = MaskedBackbone(bb, [8, 16, 32], cfg.backbone_channels)
        backbone = Joiner(d2_backbone, PositionEmbeddingSine(cfg.hidden_dim // 2, offset=-0.5))
        backbone.num_channels = d2_backbone.num_channels
        backbone.strides = d2_backbone.feature_strides
        transformer = DeformableTransformerVLDINO(cfg, self.precision)
        model = DeformableDETRDINO(backbone, transformer, cfg)
        # hipie_img.py:153: AutoTokenizer.from_pretrained("projects/HIPIE/bert-base-uncased") -- a cwd-relative asset (absent on the
        # GPU box).  The vocabulary file is all that is needed: hipie_amd.tokenizer.BertWordPiece is the same algorithm without the
        # transformers dependency (checked against transformers.BertTokenizerFast in tests/test_host_logic.py).
        from .tokenizer import BertWordPiece
        self.tokenizer = BertWordPiece.from_dir(os.environ.get("HIPIE_BERT_DIR", "projects/HIPIE/bert-base-uncased"))
        self.text_encoder = nn.Sequential(OrderedDict([("body", BertEncoder(cfg))]))
        self.detr = DDETRSegmUniDN(model, cfg, self.precision)
        self.register_buffer("pixel_mean", torch.tensor(cfg.pixel_mean).view(3, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.pixel_std).view(3, 1, 1), persistent=False)
        # MaskCLIP score fusion (hipie_img.py:248-262).  The CLIP towers are not part of a HIPIE checkpoint (the reference downloads
        # the OpenAI weights through open_clip at construction): load them with self.clip.load_clip_state_dict(); the training
        # vocabulary (hipie_img.py:72) comes from the reference installation's label file when it is present.
        self.enable_clip = bool(cfg.clip_enabled)
        if self.enable_clip:
            from .open_vocab import MaskCLIP, load_openseg_labels
            self.clip = MaskCLIP(cfg.clip_name)
            self.train_labels = load_openseg_labels("coco_panoptic", prompt_engineered=True)
        self.eval()
        # Weight-derived state (policy-dtype casts, BN folds, fused projection weights, per-geometry constants) is built by
        # finalize().  It runs lazily on the first forward and again after every load_state_dict, so the order
        # build -> DetectionCheckpointer.load -> forward of train_net.py:269-271 never sees folds of the random init.
        self._final = False
        self.register_load_state_dict_post_hook(HIPIE_IMG._after_load)

    @staticmethod
    def _after_load(module, incompatible_keys):
        # the CLIP towers are loaded separately (MaskCLIP.load_clip_state_dict): a HIPIE checkpoint never has them
        incompatible_keys.missing_keys[:] = [k for k in incompatible_keys.missing_keys if not k.startswith("clip.")]
        module._invalidate()

    def _invalidate(self):
        self._final = False

    def finalize(self):
        """after loading weights: move to the device and put GEMM weights in the policy dtypes."""
        self.to(self.device)
        self._final = True
        if self.device.type == "cuda" and os.environ.get("HIPIE_MIOPEN_FIND", "1") != "0":
            # MIOpen "find": benchmark the applicable solvers once per convolution configuration instead of the
            # immediate-mode heuristic (the shapes are static over an evaluation run); -7 ms per bs-8 ViT-H step
            torch.backends.cudnn.benchmark = True
        for m in self.modules():                       # per-geometry caches that fold (now final) parameters
            if hasattr(m, "_own_cache"):
                m._own_cache.clear()
        bb = self.detr.detr.backbone[0].backbone
        if hasattr(bb, "cast_weights"):
            bb.cast_weights()
        hd, ad = self.precision.head, self.precision.act
        cast_head(self.detr.detr.transformer.encoder, hd, ad)        # the Nv-token streams (21760 tokens / image)
        cast_head(self.detr.detr.transformer.decoder.layers, hd, ad)       # the query stream lives in `act` ...
        cast_head(self.detr.detr.transformer.decoder.ref_point_head, hd)      # ... query_pos, the box / class / IoU heads stay fp32
        cast_head(self.detr.detr.input_proj, hd, ad)
        cast_head(self.detr.mask_dino.pixel_decoder, hd, ad)
        cast_head(self.detr.mask_dino.predictor.decoder.layers, hd, ad)
        cast_head(self.detr.mask_dino.predictor.decoder.ref_point_head, hd)
        cast_head(self.detr.mask_head, hd, ad)
        self.text_encoder[0].model.set_compute_dtype(self.precision.text)
        # Precision.split3: every linear of the path on the split-fp16 GEMM (the weights stay fp32 parameters; their HL8 copies are
        # built lazily and cached per parameter version).  The flag is also cleared here, so one process can hold several policies.
        set_split(self, self.precision.split)
        self.text_encoder[0].model.split = self.precision.split
        return self

    # ---- hipie_img.py:880-898 -------------------------------------------------------------------------------
    def preprocess_image(self, batched_inputs):
        raw = [x["image"].to(self.device) for x in batched_inputs]
        if len(set(tuple(t.shape) for t in raw)) == 1:         # equal sizes: one normalisation pass over the stacked batch
            t = (torch.stack(raw).float() - self.pixel_mean) / self.pixel_std
            return ImageList(t, [(int(t.shape[-2]), int(t.shape[-1]))] * len(raw))
        images = [(t.float() - self.pixel_mean) / self.pixel_std for t in raw]
        return ImageList.from_tensors(images)

    # ---- hipie_img.py:900-922 -------------------------------------------------------------------------------
    def forward_text(self, batched_inputs, device=None):
        if "input_ids" in batched_inputs[0]:               # synthetic / pre-tokenised path (no vocab on the GPU box)
            # ids / masks stay where the caller keeps them: host tensors (what a tokenizer produces) let the text encoder trim the
            # masked tail without a device round trip (BertEncoder.forward)
            ids = torch.stack([x["input_ids"] for x in batched_inputs])
            mask = torch.stack([x["attention_mask"] for x in batched_inputs])
            sep = 1012
        else:
            if self.tokenizer is None:
                raise RuntimeError("no tokenizer assets (projects/HIPIE/bert-base-uncased): pass input_ids/attention_mask")
            captions = [x["expressions"] for x in batched_inputs]
            tok = self.tokenizer(captions, max_length=self.cfg.max_query_len, padding="max_length" if self.cfg.pad_max else "longest",
                                 return_special_tokens_mask=True, return_tensors="pt", truncation=True)   # hipie_img.py:904-909
            ids, mask = tok.input_ids, tok.attention_mask
            sep = int(self.tokenizer(".").input_ids[0, 1])                                           # bert_model.py:68-73
        return self.text_encoder[0]({"input_ids": ids, "attention_mask": mask}, sep=sep, compact=True)

    @torch.no_grad()
    def forward_raw(self, batched_inputs):
        """list of {"image": (3,h,w) RGB 0..255, "task": ..., "expressions" | "input_ids"+"attention_mask"} -> a22 dict."""
        tasks = set(x["task"] for x in batched_inputs)
        assert len(tasks) == 1
        task = tasks.pop()
        if not self._final:
            self.finalize()
        images = self.preprocess_image(batched_inputs)
        lang = self.forward_text(batched_inputs)
        outputs, _ = self.detr.coco_inference(images, None, None, train=False, language_dict_features=lang, task=task)
        outputs["image_sizes"] = images.image_sizes
        return outputs

    @torch.no_grad()
    def forward(self, batched_inputs, do_postprocess=True):
        from .postprocess import inference
        out = self.forward_raw(batched_inputs)
        return inference(self, out, batched_inputs, do_postprocess=do_postprocess)

    # ---- test / bench hooks ---------------------------------------------------------------------------------
    def pin_topk(self, topk_fg=None, topk_md=None):
        self.detr.detr.transformer.pinned_topk = topk_fg
        self.detr.mask_dino.predictor.pinned_topk = topk_md

    def last_topk(self):
        return self.detr.detr.transformer.last_topk, self.detr.mask_dino.predictor.last_topk
