"""Ground truth of one batch in the form the matchers and criteria take (hipie/hipie_img.py:422-447 ``prepare_targets``; the thing / stuff
split at the head of ``coco_forward``, hipie/models/ddetrs_dn.py:279-296)."""
import torch

_PER_OBJECT = ("labels", "boxes", "masks", "positive_map", "is_thing")


def prepare_targets(instances, device=None, half=False):
    """per image an Instances-like object with image_size (h, w), gt_classes (n,), gt_boxes (.tensor (n, 4) absolute xyxy), positive_map
    (n, L) bool, is_thing (n,) bool and optionally gt_masks (a BitMasks with .tensor, or a plain (n, h, w) tensor as the LSJ mapper leaves
    it) -> [{"labels", "boxes" (n, 4) cxcywh / image size, "image_size" (w, h, w, h), "positive_map", "is_thing"[, "masks"]}];
    half: the AMP variant (boxes, image size and masks in fp16)."""
    out = []
    for inst in instances:
        h, w = inst.image_size
        size = torch.as_tensor([w, h, w, h], dtype=torch.float, device=device if device is not None else inst.gt_classes.device)
        xyxy = inst.gt_boxes.tensor / size
        boxes = torch.cat(((xyxy[:, :2] + xyxy[:, 2:]) / 2, xyxy[:, 2:] - xyxy[:, :2]), 1)
        if half:
            boxes, size = boxes.half(), size.half()
        t = {"labels": inst.gt_classes, "boxes": boxes, "image_size": size, "positive_map": inst.positive_map, "is_thing": inst.is_thing}
        if inst.has("gt_masks"):
            m = inst.gt_masks
            m = m.tensor if hasattr(m, "tensor") else m
            t["masks"] = m.half() if half else m
        out.append(t)
    return out


def split_things_stuff(targets):
    """-> (thing targets, stuff targets): the per-object entries filtered by `is_thing`, "image_size" carried over"""
    fg, bg = [], []
    for t in targets:
        keep = t["is_thing"].bool()
        for dst, sel in ((fg, keep), (bg, ~keep)):
            d = {k: v[sel] for k, v in t.items() if k in _PER_OBJECT}
            d["image_size"] = t["image_size"]
            dst.append(d)
    return fg, bg
