"""CPU restatement of the post-processing row -- test infrastructure, see oracle/__init__.py.

Follows, per image and with the reference's own loops:
  HIPIE_IMG.inference            projects/HIPIE/hipie/hipie_img.py:537-766   (CLIP fusion off: row f-2)
  HIPIE_IMG.panoptic_inference   hipie_img.py:473-535
  HIPIE_IMG.semantic_inference   hipie_img.py:870-878
  convert_grounding_to_od_logits hipie_img.py:1025-1052
  segmentation_postprocess       hipie/models/ddetrs.py:1029-1076
  Boxes.scale / clip / nonempty  detectron2/structures/boxes.py:183-215
torchvision.ops.batched_nms is THIRD-PARTY and absent from /root/reference (no version pinned by the reference); it is
restated here from the published algorithm (torchvision/ops/boxes.py batched_nms + csrc/ops/cpu/nms_kernel.cpp).
Pinned by tests/golden/post.npz, produced by running the reference's own hipie_img.py (tests/golden/gen_golden.py post).
"""
import torch
import torch.nn.functional as F

DEFAULTS = dict(ota=True, mask_thres=0.5, mask_stride=4, transform_eval=True, pano_temp=0.06, overlap_threshold=0.8,
                object_mask_threshold=0.25, use_bg_for_pano=True, bg_cls_agnostic=False, max_pool=False,
                mode_free=False, nms_thresh=0.7, clip=None, clip_fg_a=0.3, clip_fg_b=1.7, pano_temp_fg=0.06)


def convert_grounding_to_od_logits(logits, num_classes, positive_map, is_thing={}, mode=None, model_free=False,
                                   max_pool=False):
    """(bs, Q, L) token logits -> (bs, Q, num_classes): mean (or max) over each class's token span; in FG mode stuff
    classes, in BG mode thing classes, are set to -9999 (hipie_img.py:1025-1052)."""
    if model_free:
        mode = None
    scores = torch.zeros(logits.shape[0], logits.shape[1], num_classes)
    for label_j in positive_map:
        tok = logits[:, :, torch.as_tensor(positive_map[label_j], dtype=torch.long)]
        scores[:, :, label_j - 1] = tok.max(-1)[0] if max_pool else tok.mean(-1)
        if mode == "FG" and not is_thing.get(label_j, True):
            scores[:, :, label_j - 1] = -9999.0
        elif mode == "BG" and is_thing.get(label_j, True):
            scores[:, :, label_j - 1] = -9999.0
    return scores


def box_cxcywh_to_xyxy(x):
    """hipie/util/box_ops.py:17-21."""
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


def nms(boxes, scores, thr):
    """greedy NMS: stable descending sort; IoU = inter / (area_i + area_j - inter) > thr suppresses."""
    x1, y1, x2, y2 = [t.tolist() for t in boxes.unbind(1)]
    f32 = torch.float32
    areas = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))
    order = scores.sort(dim=0, descending=True, stable=True)[1].tolist()
    n = len(order)
    dead = [False] * n
    keep = []
    b = boxes
    for a in range(n):
        i = order[a]
        if dead[i]:
            continue
        keep.append(i)
        rest = torch.tensor([j for j in order[a + 1:] if not dead[j]], dtype=torch.long)
        if rest.numel() == 0:
            continue
        w = (torch.minimum(b[i, 2], b[rest, 2]) - torch.maximum(b[i, 0], b[rest, 0])).clamp(min=0)
        h = (torch.minimum(b[i, 3], b[rest, 3]) - torch.maximum(b[i, 1], b[rest, 1])).clamp(min=0)
        inter = (w * h).to(f32)
        ovr = inter / (areas[i] + areas[rest] - inter)
        for j in rest[ovr > thr].tolist():
            dead[j] = True
    return torch.tensor(keep, dtype=torch.long)


def batched_nms(boxes, scores, idxs, thr):
    """torchvision.ops.batched_nms: <= 4000 coordinates -> coordinate trick, else per-class NMS."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for c in torch.unique(idxs):
            ci = torch.where(idxs == c)[0]
            keep_mask[ci[nms(boxes[ci], scores[ci], thr)]] = True
        k = torch.where(keep_mask)[0]
        return k[scores[k].sort(descending=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return nms(boxes + offsets[:, None], scores, thr)


def panoptic_inference(mask_cls, mask_pred, is_thing, object_mask_threshold, overlap_threshold):
    """hipie_img.py:473-535.  mask_cls (N, C) probabilities, mask_pred (N, H, W) logits."""
    scores, labels = mask_cls.max(-1)
    mask_pred = mask_pred.sigmoid()
    keep = scores > object_mask_threshold
    cur_scores, cur_classes, cur_masks = scores[keep], labels[keep], mask_pred[keep]
    cur_prob_masks = cur_scores.view(-1, 1, 1) * cur_masks
    h, w = cur_masks.shape[-2:]
    panoptic_seg = torch.zeros((h, w), dtype=torch.int32)
    segments_info = []
    current_segment_id = 0
    if cur_masks.shape[0] == 0:
        return panoptic_seg, segments_info
    cur_mask_ids = cur_prob_masks.argmax(0)
    stuff_memory_list = {}
    for k in range(cur_classes.shape[0]):
        pred_class = int(cur_classes[k])
        isthing = is_thing.get(pred_class + 1, True)
        mask_area = int((cur_mask_ids == k).sum())
        original_area = int((cur_masks[k] >= 0.5).sum())
        mask = (cur_mask_ids == k) & (cur_masks[k] >= 0.5)
        if mask_area > 0 and original_area > 0 and int(mask.sum()) > 0:
            if mask_area / original_area < overlap_threshold:
                continue
            if not isthing:
                if pred_class in stuff_memory_list:
                    panoptic_seg[mask] = stuff_memory_list[pred_class]
                    continue
                stuff_memory_list[pred_class] = current_segment_id + 1
            current_segment_id += 1
            panoptic_seg[mask] = current_segment_id
            segments_info.append({"id": current_segment_id, "isthing": bool(isthing), "category_id": pred_class})
    return panoptic_seg, segments_info


def semantic_inference(mask_cls, mask_pred):
    """hipie_img.py:870-878."""
    return torch.einsum("qc,qhw->chw", mask_cls, mask_pred.sigmoid())


def segmentation_postprocess(inst, out_h, out_w):
    """ddetrs.py:1029-1076 on a dict(image_size, boxes (n,4) xyxy, scores, classes, masks (n,1,h,w) bool)."""
    if inst["boxes"].shape[0] == 0:
        return inst
    sx, sy = out_w / inst["image_size"][1], out_h / inst["image_size"][0]
    boxes = inst["boxes"].clone()
    boxes[:, 0::2] *= sx
    boxes[:, 1::2] *= sy
    boxes[:, 0].clamp_(min=0, max=out_w)
    boxes[:, 1].clamp_(min=0, max=out_h)
    boxes[:, 2].clamp_(min=0, max=out_w)
    boxes[:, 3].clamp_(min=0, max=out_h)
    ne = ((boxes[:, 2] - boxes[:, 0]) > 0) & ((boxes[:, 3] - boxes[:, 1]) > 0)
    masks = F.interpolate(inst["masks"][ne].float(), size=(out_h, out_w), mode="nearest").squeeze(1).byte()
    return dict(image_size=(out_h, out_w), boxes=boxes[ne], scores=inst["scores"][ne], classes=inst["classes"][ne], masks=masks)


def inference(a22, image_sizes, positive_map, task, is_thing, out_sizes=None, num_bg=10, **kw):
    """HIPIE_IMG.inference (hipie_img.py:537-766) followed by segmentation_postprocess (hipie_img.py:356-362), decouple_decoder
    True, bg_query_from_lang False, demo_only False, score_thres 0.  Returns a list of dicts:
    instances (dict of tensors), panoptic_seg ((H,W) int32, segments_info), sem_seg (C,H,W).
    kw["clip"]: MODEL.CLIP.ENABLED -- a callable (i, mask_logits (Q,h,w), pred_open_prob (Q,C)) -> fused class logits (Q,C)
    (HIPIE_IMG.get_clip_logits, oracle/clip.py) used at the two call sites hipie_img.py:592-609 and :735-747."""
    o = dict(DEFAULTS)
    o.update(kw)
    max_num_inst = {"detection": 100, "grounding": 1}[task]
    num_classes = len(positive_map)
    out_sizes = out_sizes or image_sizes
    box_cls, box_pred = a22["pred_logits"][:, num_bg:], a22["pred_boxes"][:, num_bg:]
    mask_pred, iou_pred = a22["pred_masks"][:, num_bg:], a22["pred_boxious"][:, num_bg:]
    box_cls_bg, mask_pred_bg = a22["pred_logits_maskdino"], a22["pred_masks_maskdino"].unsqueeze(2)
    s = o["mask_stride"]
    results = []
    for i, image_size in enumerate(image_sizes):
        it = is_thing[i]
        has_thing = any(it.values())
        logits = convert_grounding_to_od_logits(box_cls[i][None], num_classes, positive_map, is_thing=it,
                                                mode="FG" if has_thing else None, model_free=o["mode_free"],
                                                max_pool=o["max_pool"])[0]
        if o.get("clip") is not None:
            allowed = (~(logits[:1] == -9999.0)).float()
            p_det = F.softmax(logits.sigmoid() / o["pano_temp_fg"], dim=-1) if (o["transform_eval"] and logits.shape[-1] > 1) else logits.sigmoid()
            prob = o["clip"](i, mask_pred[i][:, 0], p_det).sigmoid() * allowed
            prob = torch.sqrt((prob ** o["clip_fg_a"]) * (iou_pred[i].sigmoid() ** o["clip_fg_b"]))
        else:
            prob = torch.sqrt(logits.sigmoid() * iou_pred[i].sigmoid())
        nms_scores, idxs = torch.max(prob, 1)
        keep = batched_nms(box_cxcywh_to_xyxy(box_pred[i]), nms_scores, idxs, o["nms_thresh"])
        prob = prob[keep]
        num_inst = min(max_num_inst, prob.numel())
        boxes_i, masks_i = box_pred[i][keep], mask_pred[i][keep]
        top_v, top_i = torch.topk(prob.view(-1), num_inst, dim=0)
        top_q = torch.div(top_i, logits.shape[1], rounding_mode="floor")
        labels = top_i % logits.shape[1]
        boxes_i, masks_i = boxes_i[top_q], masks_i[top_q]
        xyxy = box_cxcywh_to_xyxy(boxes_i)
        xyxy[:, 0::2] *= image_size[1]
        xyxy[:, 1::2] *= image_size[0]
        N, C, H, W = masks_i.shape
        m = F.interpolate(masks_i, size=(H * s, W * s), mode="bilinear", align_corners=False)
        m = (m.sigmoid() > o["mask_thres"])[:, :, :image_size[0], :image_size[1]]
        inst = dict(image_size=tuple(image_size), boxes=xyxy, scores=top_v, classes=labels, masks=m)
        pan, seg_info, sem = None, None, None
        if task == "detection":
            mode = None if (o["use_bg_for_pano"] or o["bg_cls_agnostic"]) else "BG"
            logits_bg = convert_grounding_to_od_logits(box_cls_bg[i][None], num_classes, positive_map, is_thing=it, mode=mode,
                                                       model_free=o["mode_free"], max_pool=o["max_pool"])[0]
            if o["use_bg_for_pano"]:
                logits_all, masks_all = logits_bg, mask_pred_bg[i]
            else:
                logits_all = torch.cat([logits[keep], logits_bg], 0)
                masks_all = torch.cat([mask_pred[i][keep], mask_pred_bg[i]], 0)
            N, C, H, W = masks_all.shape
            if o["transform_eval"]:
                cls_all = F.softmax(logits_all.sigmoid() / o["pano_temp"], dim=-1)
            else:
                cls_all = logits_all.sigmoid()
            masks_all = F.interpolate(masks_all, size=(H * s, W * s), mode="bilinear", align_corners=False)
            masks_all = masks_all[:, :, :image_size[0], :image_size[1]]
            if o.get("clip") is not None:
                cls_all = o["clip"](i, masks_all[:, 0], cls_all).softmax(-1)
            up = F.interpolate(masks_all, size=tuple(out_sizes[i]), mode="bilinear", align_corners=False)[:, 0]
            sem = semantic_inference(cls_all, up)
            pan, seg_info = panoptic_inference(cls_all, up, it, o["object_mask_threshold"], o["overlap_threshold"])
        inst = segmentation_postprocess(inst, out_sizes[i][0], out_sizes[i][1])
        results.append(dict(instances=inst, panoptic_seg=(pan, seg_info), sem_seg=sem))
    return results
