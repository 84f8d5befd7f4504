"""Matching costs and the assignment, for both heads of the reference:

* the detection branch's ``HungarianMatcherVL`` (hipie/models/deformable_detr/matcher.py:317-729): token-level focal class cost against
  the target's positive map, L1 + GIoU box costs, optionally the point-sampled mask costs, "stuff" targets (no box) take the mean box
  cost of the "thing" targets; a second entry point (``force_box_loss``, :624-729) used for the encoder proposals: class + box costs
  only, no stuff handling;
* MaskDINO's ``HungarianMatcher`` (hipie/models/maskdino/matcher.py:76-259): the same costs with class ids or positive maps.

One cost routine serves all of them.  The cost matrix is built on the tensors' device (one image at a time, like the reference); the
linear sum assignment itself runs on the host in scipy, as in the reference (``C.cpu()`` -> ``linear_sum_assignment``)."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn

from .boxes import box_cxcywh_to_xyxy, generalized_box_iou


@dataclass
class MatchWeights:
    cls: float = 1.0
    l1: float = 1.0
    giou: float = 1.0
    mask: float = 1.0
    dice: float = 1.0


def point_sample(maps, coords):
    """maps (N, C, H, W), coords (N, P, 2) in [0, 1]^2 as (x, y) -> (N, C, P): bilinear, zero outside, pixel centres at (i + 0.5) / size
    (detectron2 point_rend point_features.py:19-42 with align_corners=False)."""
    return F.grid_sample(maps, 2.0 * coords[:, :, None, :] - 1.0, align_corners=False).squeeze(3)


def focal_token_cost(prob, alpha=0.25, gamma=2.0):
    """per (query, token): focal cost of calling the token positive minus the cost of calling it negative (matcher.py:511-527)."""
    neg = (1 - alpha) * prob ** gamma * (-(1 - prob + 1e-8).log())
    pos = alpha * (1 - prob) ** gamma * (-(prob + 1e-8).log())
    return pos - neg


def class_cost(prob, target, mode="auto"):
    """prob (Q, L) sigmoid scores.  mode "map" (or "auto" with a "positive_map" (T, L) bool in the target): the mean of the token cost
    over the target's positive tokens (a class name may be several tokens; a (T, 1) all-true map is the encoder's binary objectness).
    mode "ids" (or "auto" without a map): "labels" (T,) index the columns (maskdino/matcher.py:208-217)."""
    tok = focal_token_cost(prob)
    pm = target.get("positive_map") if mode != "ids" else None
    if pm is None:
        if mode == "map":
            raise KeyError("class_cost: target has no positive_map")
        return tok[:, target["labels"]]
    pm = pm.to(tok.dtype)
    return (tok @ pm.t()) / pm.sum(1)[None, :]          # no positive token: 0 / 0 = NaN, as the reference's mean over nothing


def mask_costs(pred, tgt, coords):
    """pred (Q, H, W) logits, tgt (T, H', W') {0, 1}, coords (P, 2): the SAME points for every mask (matcher.py:567-597) ->
    sigmoid-CE cost and dice cost, both (Q, T) (matcher.py:22-69)."""
    P = coords.shape[0]
    x = point_sample(pred[:, None].float(), coords[None].expand(pred.shape[0], P, 2))[:, 0]
    t = point_sample(tgt[:, None].float(), coords[None].expand(tgt.shape[0], P, 2))[:, 0]
    ce = (F.softplus(-x) @ t.t() + F.softplus(x) @ (1 - t).t()) / P
    s = x.sigmoid()
    dice = 1 - (2 * (s @ t.t()) + 1) / (s.sum(-1)[:, None] + t.sum(-1)[None, :] + 1)
    return ce, dice


def cost_matrix(logits, boxes, target, w, masks=None, coords=None, stuff_takes_mean=True, with_boxes=True, class_mode="auto"):
    """One image: logits (Q, L), boxes (Q, 4) cxcywh, target dict (boxes (T, 4), positive_map | labels, is_thing (T,), masks) -> (Q, T)."""
    prob = logits.sigmoid()
    C = w.cls * class_cost(prob, target, class_mode)
    if with_boxes:
        tb = target["boxes"]
        l1 = torch.cdist(boxes, tb, p=1)
        gi = -generalized_box_iou(box_cxcywh_to_xyxy(boxes), box_cxcywh_to_xyxy(tb))
        if stuff_takes_mean:
            thing = target["is_thing"].bool()
            # an image without things: the mean over nothing is NaN, which the reference zeroes together with every other NaN
            l1[:, ~thing] = l1[:, thing].mean()
            gi[:, ~thing] = gi[:, thing].mean()
            l1 = torch.nan_to_num(l1, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
            gi = torch.nan_to_num(gi, nan=0.0, posinf=float("inf"), neginf=float("-inf"))
        C = C + w.l1 * l1 + w.giou * gi
    if masks is not None:
        ce, dice = mask_costs(masks, target["masks"].to(masks), coords)
        C = C + w.mask * ce + w.dice * dice
    return C


def assign(C):
    """(Q, T) cost -> (query indices, target indices) int64 on the host, rows in increasing order (scipy's contract)."""
    from scipy.optimize import linear_sum_assignment
    i, j = linear_sum_assignment(C.detach().cpu().numpy())
    return torch.as_tensor(i, dtype=torch.int64), torch.as_tensor(j, dtype=torch.int64)


class HungarianMatcher(nn.Module):
    """weights: MatchWeights.  num_points: points of the mask costs (12544 = 112^2 in both heads).  stuff_takes_mean: the panoptic
    box handling (`panoptic_box_loss` / `panoptic_on`).  draw(shape, device) supplies uniform [0, 1) numbers (default torch.rand): the
    reference draws ONE (1, num_points, 2) tensor per image, in batch order."""

    def __init__(self, weights=None, num_points=112 * 112, stuff_takes_mean=True, draw=None, class_mode="auto"):
        super().__init__()
        self.class_mode = class_mode                       # "map": positive maps (vl_loss), "ids": class ids, "auto": whichever the target has
        self.w = weights or MatchWeights()
        if not any((self.w.cls, self.w.l1, self.w.giou, self.w.mask)):
            raise ValueError("all matching costs are zero")
        self.num_points = num_points
        self.stuff_takes_mean = stuff_takes_mean
        self.draw = draw or (lambda shape, device: torch.rand(shape, device=device))

    @torch.no_grad()
    def forward(self, logits, boxes, targets, masks=None, costs=("cls", "box", "mask")):
        """logits (B, Q, L), boxes (B, Q, 4), masks None | per-image sequence of (Q, H, W) -> [(query idx, target idx)] per image."""
        out = []
        for b, tgt in enumerate(targets):
            m = masks[b] if (masks is not None and "mask" in costs) else None
            coords = self.draw((1, self.num_points, 2), logits.device)[0] if m is not None else None
            C = cost_matrix(logits[b], boxes[b], tgt, self.w, m, coords, self.stuff_takes_mean, "box" in costs, self.class_mode)
            out.append(assign(C.reshape(logits.shape[1], -1)))
        return out

    @torch.no_grad()
    def forward_boxes_only(self, logits, boxes, targets):
        """the `force_box_loss` entry (matcher.py:640-729): class + L1 + GIoU, stuff targets keep their own box costs."""
        return [assign(cost_matrix(logits[b], boxes[b], t, self.w, stuff_takes_mean=False, class_mode=self.class_mode)) for b, t in enumerate(targets)]
