"""The set-prediction losses of the two heads.

``DetCriterion``  -- the detection branch: ``SetCriterion`` + ``DINOCriterion`` of hipie/models/deformable_detr/deformable_detr.py:311-854
                     (token-level focal loss against the positive maps, L1 / GIoU / IoU-score box losses with the panoptic "stuff has no
                     box" weighting, point-sampled mask losses, auxiliary layers, the binary encoder-proposal loss, de-noising losses).
``MaskCriterion`` -- MaskDINO: ``SetCriterion`` of hipie/models/maskdino/criterion.py:129-465 (its own matching per output, class-id or
                     positive-map focal loss, sigmoid-CE + dice on importance-sampled points, de-noising and intermediate outputs).

What is NOT restated: the BoxInst projection / pairwise-colour terms (`loss_masks_boxinst`, :526-596, behind MODEL.BOXINST.ENABLED, off
in every shipped config) and the tracking `loss_reid` (:598-634, video models: SURVEY section 8 "out").  With OTA matching (matcher.forward_ota)
the reference normalises by the number of matched pairs instead of targets (`num_boxes = len(idx[0]) if self.ota`): pass ota=True.
The cross-rank mean of the target count (`all_reduce(num_boxes) / world_size`) is applied when torch.distributed is initialised, as
in the reference.  Random point coordinates come from ``draw(shape, device)`` (default torch.rand), in the reference's order."""
import copy

import torch
import torch.nn.functional as F
from torch import nn

from .boxes import box_cxcywh_to_xyxy, generalized_box_iou, paired_giou_loss, paired_iou
from .dn import dn_match_indices
from .matcher import point_sample


# ---- elementwise losses ---------------------------------------------------------------------------------------------------------
def _focal(logits, targets, alpha, gamma):
    p = logits.sigmoid()
    ce = F.binary_cross_entropy_with_logits(logits, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss


def token_focal_loss(logits, onehot, text_mask=None, alpha=0.25, gamma=2.0):
    """sum of the binary focal loss over (image, query, token), pad tokens (text_mask == 0) left out
    (token_sigmoid_binary_focal_loss, deformable_detr/segmentation.py:120-166)."""
    if text_mask is not None:
        keep = (text_mask > 0)[:, None, :].expand_as(logits)
        logits, onehot = logits[keep], onehot[keep]
    return _focal(logits, onehot, alpha, gamma).sum()


def dense_focal_loss(logits, targets, count, alpha=0.25, gamma=2.0):
    """mean over axis 1, summed, per `count` (segmentation.py:92-117; maskdino/criterion.py:28-54)."""
    return _focal(logits, targets, alpha, gamma).mean(1).sum() / count


def dice_loss(logits, targets, count):
    """(jit_loss.py:28-48)"""
    s = logits.sigmoid().flatten(1)
    return (1 - (2 * (s * targets).sum(-1) + 1) / (s.sum(-1) + targets.sum(-1) + 1)).sum() / count


def sigmoid_ce_loss(logits, targets, count):
    """(jit_loss.py:4-22)"""
    return F.binary_cross_entropy_with_logits(logits, targets, reduction="none").mean(1).sum() / count


def uncertain_points(logits, num_points, oversample, importance, draw):
    """logits (N, 1, H, W) -> (N, num_points, 2): of `oversample * num_points` uniform candidates the `importance * num_points` whose
    sampled logit is closest to 0, then fresh uniform points for the rest (PointRend's get_uncertain_point_coords_with_randomness,
    detectron2 point_rend/point_features.py:63-116, with the reference's uncertainty -|logit|)."""
    N = logits.shape[0]
    cand = draw((N, int(num_points * oversample), 2), logits.device)
    score = -point_sample(logits, cand)[:, 0].abs()
    n_imp = int(importance * num_points)
    top = score.topk(n_imp, dim=1)[1]
    pts = torch.gather(cand, 1, top[:, :, None].expand(N, n_imp, 2))
    if num_points - n_imp > 0:
        pts = torch.cat((pts, draw((N, num_points - n_imp, 2), logits.device)), 1)
    return pts


def _perm(indices, which):
    return (torch.cat([torch.full_like(p[which], i) for i, p in enumerate(indices)]), torch.cat([p[which] for p in indices]))


def _target_count(targets, device):
    count = float(sum(len(t["labels"]) for t in targets))
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return max(count, 1.0)                      # one process: no device round trip (a host wait for the whole queue)
    n = torch.as_tensor([count], device=device)
    torch.distributed.all_reduce(n)
    return torch.clamp(n / torch.distributed.get_world_size(), min=1).item()


def _positive_onehot(logits, targets, indices):
    onehot = torch.zeros_like(logits)
    for b, (src, tgt) in enumerate(indices):
        if len(src):
            onehot[b, src] = targets[b]["positive_map"][tgt].to(logits.dtype)
    return onehot


def _pad_masks(masks, divisor=0):
    """per-image (n_i, h_i, w_i) masks -> (B, max n, H, W) zero padded at the bottom / right (nested_tensor_from_tensor_list), H and W rounded
    up to a multiple of `divisor`"""
    n = max(m.shape[0] for m in masks)
    H = max(m.shape[-2] for m in masks)
    W = max(m.shape[-1] for m in masks)
    if divisor > 1:
        H, W = -(-H // divisor) * divisor, -(-W // divisor) * divisor
    out = masks[0].new_zeros(len(masks), n, H, W)
    for i, m in enumerate(masks):
        out[i, :m.shape[0], :m.shape[-2], :m.shape[-1]] = m
    return out


# ---- detection branch ---------------------------------------------------------------------------------------------------------------
class DetCriterion(nn.Module):
    """losses: any of "labelsVL", "boxes", "masks".  matcher: only its ``forward_boxes_only`` is called here (the encoder proposals,
    deformable_detr.py:693-716); the assignments of the decoder layers are computed by the caller (ddetrs_dn.py coco_forward) and
    passed as `indices_per_layer` (last entry = last layer)."""

    def __init__(self, matcher, losses, focal_alpha=0.25, mask_out_stride=4, point_sample_masks=True, panoptic_box_loss=True,
                 still_cls_for_encoder=False, num_points=112 * 112, oversample_ratio=3.0, importance_sample_ratio=0.75, draw=None, ota=False):
        super().__init__()
        self.ota = ota                                     # one-to-many assignments: every loss is per MATCHED PAIR (deformable_detr.py:362, 430, 481)
        self.matcher, self.losses = matcher, tuple(losses)
        self.focal_alpha, self.mask_out_stride = focal_alpha, mask_out_stride
        self.point_sample_masks, self.panoptic_box_loss = point_sample_masks, panoptic_box_loss
        self.still_cls_for_encoder = still_cls_for_encoder
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = num_points, oversample_ratio, importance_sample_ratio
        self.draw = draw or (lambda shape, device: torch.rand(shape, device=device))

    # -- the three loss families (each: outputs of ONE layer, its assignment, the normaliser) --
    def loss_labels(self, out, targets, indices, count):
        """deformable_detr.py:353-381"""
        logits = out["pred_logits"]
        if self.ota:
            count = sum(len(p[0]) for p in indices)
        if count == 0:
            return {"loss_ce": logits.sum() * 0.0}
        onehot = _positive_onehot(logits, targets, indices)
        return {"loss_ce": token_focal_loss(logits, onehot, out["text_masks"], self.focal_alpha) / count}

    def loss_boxes(self, out, targets, indices, count):
        """deformable_detr.py:397-450: stuff targets carry no box -- they are masked out and the rest re-weighted to the full count"""
        bi, si = _perm(indices, 0)
        src = out["pred_boxes"][bi, si]
        tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
        thing = torch.cat([t["is_thing"][j] for t, (_, j) in zip(targets, indices)]).float().reshape(-1, 1)
        if not self.panoptic_box_loss:
            thing = torch.ones_like(thing)
        if len(tgt) == 0 or thing.sum() == 0:
            return {"loss_bbox": src.sum() * 0.0, "loss_giou": src.sum() * 0.0}
        reweight = thing.shape[0] / (thing.sum() + 1e-6)
        if self.ota:
            count = src.shape[0]
        sx, tx = box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)
        res = {"loss_bbox": (F.l1_loss(src, tgt, reduction="none") * thing * reweight).sum() / count,
               "loss_giou": (paired_giou_loss(sx, tx) * thing[:, 0] * reweight).sum() / count}
        if "pred_boxious" in out:
            with torch.no_grad():
                iou = paired_iou(sx, tx)
            score = out["pred_boxious"][bi, si].flatten()
            res["loss_boxiou"] = (F.binary_cross_entropy_with_logits(score, iou, reduction="none") * thing[:, 0]).mean() * reweight
        return res

    def target_masks(self, targets, like):
        """ground-truth masks padded to /32 and sub-sampled at the mask stride from its centre pixel (deformable_detr.py:725-737)"""
        t = _pad_masks([x["masks"] for x in targets], 32).to(like)
        s = self.mask_out_stride
        if s != 1:
            H, W = t.shape[-2:]
            t = t[..., s // 2::s, s // 2::s]
            if t.shape[-2] * s != H or t.shape[-1] * s != W:
                raise ValueError("mask size is not a multiple of the mask stride")
        return t

    def loss_masks(self, out, targets, indices, count):
        """deformable_detr.py:452-524.  pred_masks: a list of per-image (1, n_i, frames, h, w) tensors holding the MATCHED instances in
        assignment order (forward_mask_head_train), or one such tensor."""
        src = out["pred_masks"]
        if isinstance(src, (list, tuple)):
            src = torch.cat(list(src), 1)[0]
        if src.ndim == 0:                                   # box-only supervision
            return {"loss_mask": src * 0.0, "loss_dice": src * 0.0}
        frames = src.shape[1]
        tm = self.target_masks(targets, src)
        tm = tm.reshape(len(targets), -1, frames, tm.shape[-2], tm.shape[-1])[_perm(indices, 1)]
        if self.ota:
            count = src.shape[0]
        if len(tm) == 0:
            return {"loss_mask": src.sum() * 0.0, "loss_dice": src.sum() * 0.0}
        if self.point_sample_masks:
            with torch.no_grad():
                pts = uncertain_points(src, self.num_points, self.oversample_ratio, self.importance_sample_ratio, self.draw)
                lab = point_sample(tm, pts)[:, 0]
            lg = point_sample(src, pts)[:, 0]
            return {"loss_mask": dense_focal_loss(lg, lab, count), "loss_dice": dice_loss(lg, lab, count)}
        return {"loss_mask": dense_focal_loss(src.flatten(1), tm.flatten(1), count), "loss_dice": dice_loss(src.flatten(1), tm.flatten(1), count)}

    def _one(self, name, out, targets, indices, count):
        return {"labelsVL": self.loss_labels, "boxes": self.loss_boxes, "masks": self.loss_masks}[name](out, targets, indices, count)

    def forward(self, outputs, targets, indices_per_layer, dn_meta=None):
        """outputs: last layer's pred_* (+ "text_masks"), optional "aux_outputs" [layer dicts], "enc_outputs"; dn_meta: the dict of
        ``cdn_queries`` extended by the caller with "output_known_lbs_bboxes" = the de-noising part's outputs (+ its "aux_outputs").
        -> {name: scalar} with the reference's key suffixes (_i aux layer, _enc, _dn, _dn_i)."""
        count = _target_count(targets, outputs["pred_logits"].device)
        losses = {}
        for name in self.losses:
            losses.update(self._one(name, outputs, targets, indices_per_layer[-1], count))
        for i, aux in enumerate(outputs.get("aux_outputs", ())):
            for name in self.losses:
                losses.update({k + "_%d" % i: v for k, v in self._one(name, aux, targets, indices_per_layer[i], count).items()})
        if "enc_outputs" in outputs:                       # binary objectness of the encoder proposals: every target is class 0
            enc = dict(outputs["enc_outputs"])
            bin_targets = copy.deepcopy(targets)
            for t in bin_targets:
                t["labels"] = torch.zeros_like(t["labels"])
                if self.still_cls_for_encoder and "positive_map" in t:
                    t["positive_map"] = torch.ones(len(t["positive_map"]), 1, dtype=torch.bool, device=t["positive_map"].device)
                    enc["text_masks"] = None
            idx = self.matcher.forward_boxes_only(enc["pred_logits"], enc["pred_boxes"], bin_targets)
            for name in self.losses:
                if name != "masks":
                    losses.update({k + "_enc": v for k, v in self._one(name, enc, bin_targets, idx, count).items()})
        losses.update(self.dn_losses(dn_meta, targets, len(outputs.get("aux_outputs", ())), count, outputs["pred_logits"].device))
        return losses

    def dn_losses(self, dn_meta, targets, n_aux, count, device):
        """deformable_detr.py:774-853: label and box losses of the de-noising queries against their own targets, per `count * groups`"""
        names = [n for n in self.losses if n in ("labelsVL", "boxes")]
        out = {}
        known = dn_meta.get("output_known_lbs_bboxes") if dn_meta else None
        if known is None:
            zero = torch.zeros((), device=device)
            for sfx in [""] + ["_%d" % i for i in range(n_aux)]:
                out.update({"loss_bbox_dn" + sfx: zero, "loss_giou_dn" + sfx: zero, "loss_class_dn" + sfx: zero})
            return out
        idx = dn_match_indices(targets, {"dn_num": dn_meta["dn_num"], "single_padding": dn_meta["single_padding"]}, device)
        for name in names:
            out.update({k + "_dn": v for k, v in self._one(name, known, targets, idx, count * dn_meta["dn_num"]).items()})
        for i in range(n_aux):
            for name in names:
                out.update({k + "_dn_%d" % i: v for k, v in self._one(name, known["aux_outputs"][i], targets, idx, count * dn_meta["dn_num"]).items()})
        return out


# ---- MaskDINO ------------------------------------------------------------------------------------------------------------------------
class MaskCriterion(nn.Module):
    """losses: any of "labels", "masks", "boxes"; vl_loss: labels against positive maps (token focal loss) instead of class ids;
    dn: "no" | "standard" | "seg" with dn_losses the families applied to the de-noising part (maskdino/criterion.py:136-166)."""

    def __init__(self, num_classes, matcher, losses, vl_loss=False, num_points=112 * 112, oversample_ratio=3.0, importance_sample_ratio=0.75,
                 dn="no", dn_losses=(), panoptic_on=False, focal_alpha=0.25, draw=None):
        super().__init__()
        self.num_classes, self.matcher, self.losses = num_classes, matcher, tuple(losses)
        self.vl_loss, self.dn, self.dn_losses_names, self.panoptic_on, self.focal_alpha = vl_loss, dn, tuple(dn_losses), panoptic_on, focal_alpha
        self.num_points, self.oversample_ratio, self.importance_sample_ratio = num_points, oversample_ratio, importance_sample_ratio
        self.draw = draw or (lambda shape, device: torch.rand(shape, device=device))

    def loss_labels(self, out, targets, indices, count):
        logits = out["pred_logits"]
        if self.vl_loss:                                   # maskdino/criterion.py:209-238
            if count == 0:
                return {"loss_ce": logits.sum() * 0.0}
            onehot = _positive_onehot(logits, targets, indices).detach()
            return {"loss_ce": token_focal_loss(logits, onehot, out["text_masks"].detach(), self.focal_alpha) / count}
        onehot = torch.zeros_like(logits)                  # :186-207: unmatched queries are all-zero rows ("no object")
        bi, si = _perm(indices, 0)
        cls = torch.cat([t["labels"][j] for t, (_, j) in zip(targets, indices)])
        onehot[bi, si, cls] = 1
        return {"loss_ce": dense_focal_loss(logits, onehot, count, self.focal_alpha)}

    def loss_boxes(self, out, targets, indices, count):
        """:240-284: L1 and 1 - GIoU of the matched pairs; panoptic: things only"""
        bi, si = _perm(indices, 0)
        src = out["pred_boxes"][bi, si]
        tgt = torch.cat([t["boxes"][j] for t, (_, j) in zip(targets, indices)])
        if self.panoptic_on:
            thing = torch.cat([t["is_thing"][j] for t, (_, j) in zip(targets, indices)]).bool()
            src, tgt = src[thing], tgt[thing]
        giou = torch.diagonal(generalized_box_iou(box_cxcywh_to_xyxy(src), box_cxcywh_to_xyxy(tgt)))
        return {"loss_bbox": F.l1_loss(src, tgt, reduction="none").sum() / count, "loss_giou": (1 - giou).sum() / count}

    def loss_masks(self, out, targets, indices, count):
        """:286-336"""
        src = out["pred_masks"][_perm(indices, 0)][:, None]
        tm = _pad_masks([t["masks"] for t in targets]).to(src)[_perm(indices, 1)][:, None]
        with torch.no_grad():
            pts = uncertain_points(src, self.num_points, self.oversample_ratio, self.importance_sample_ratio, self.draw)
            lab = point_sample(tm, pts)[:, 0]
        lg = point_sample(src, pts)[:, 0]
        return {"loss_mask": sigmoid_ce_loss(lg, lab, count), "loss_dice": dice_loss(lg, lab, count)}

    def _one(self, name, out, targets, indices, count):
        return {"labels": self.loss_labels, "masks": self.loss_masks, "boxes": self.loss_boxes}[name](out, targets, indices, count)

    def _match(self, out, targets):
        return self.matcher(out["pred_logits"], out["pred_boxes"], targets, masks=out.get("pred_masks"))

    def forward(self, outputs, targets, mask_dict=None):
        """:370-463.  mask_dict (de-noising): {"output_known_lbs_bboxes": outputs of the de-noising part (+ "aux_outputs"), "scalar":
        groups, "pad_size": de-noising queries per image}."""
        dev = outputs["pred_logits"].device
        main = {k: v for k, v in outputs.items() if k != "aux_outputs"}
        use_dn = self.dn != "no" and mask_dict is not None
        if use_dn:
            known, groups = mask_dict["output_known_lbs_bboxes"], mask_dict["scalar"]
            if mask_dict["pad_size"] % groups:
                raise ValueError("pad_size is not a multiple of the number of de-noising groups")
            dn_idx = dn_match_indices(targets, {"dn_num": groups, "single_padding": mask_dict["pad_size"] // groups}, dev)
        idx = self._match(main, targets)
        count = _target_count(targets, dev)
        losses = {}
        for name in self.losses:
            losses.update(self._one(name, outputs, targets, idx, count))

        def dn_part(out, sfx):
            if use_dn:
                for name in self.dn_losses_names:
                    losses.update({k + "_dn" + sfx: v for k, v in self._one(name, out, targets, dn_idx, count * groups).items()})
            elif self.dn != "no":
                zero = torch.zeros((), device=dev)
                for k in ("loss_bbox", "loss_giou", "loss_ce") + (("loss_mask", "loss_dice") if self.dn == "seg" else ()):
                    losses[k + "_dn" + sfx] = zero

        dn_part(known if use_dn else None, "")
        first = 0 if "interm_outputs" in outputs else 1
        for i, aux in enumerate(outputs.get("aux_outputs", ())):
            aidx = self._match(aux, targets)
            for name in self.losses:
                losses.update({k + "_%d" % i: v for k, v in self._one(name, aux, targets, aidx, count).items()})
            if i >= first:
                dn_part(known["aux_outputs"][i] if use_dn else None, "_%d" % i)
        if "interm_outputs" in outputs:
            iidx = self._match(outputs["interm_outputs"], targets)
            for name in self.losses:
                losses.update({k + "_interm": v for k, v in self._one(name, outputs["interm_outputs"], targets, iidx, count).items()})
        return losses
