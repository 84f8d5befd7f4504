"""Matching costs and the assignment, for both heads of the reference:

* the detection branch's ``HungarianMatcherVL`` (hipie/models/deformable_detr/matcher.py:317-729): token-level focal class cost against
  the target's positive map, L1 + GIoU box costs, optionally the point-sampled mask costs, "stuff" targets (no box) take the mean box
  cost of the "thing" targets; a second entry point (``force_box_loss``, :624-729) used for the encoder proposals: class + box costs
  only, no stuff handling; and the one-to-many SimOTA matching the shipped configs train with (``forward_ota``, :347-509);
* MaskDINO's ``HungarianMatcher`` (hipie/models/maskdino/matcher.py:76-259): the same costs with class ids or positive maps.

One cost routine serves all of them.  The cost matrix is built on the tensors' device (one image at a time, like the reference); the
linear sum assignment itself runs on the host in scipy, as in the reference (``C.cpu()`` -> ``linear_sum_assignment``)."""
from dataclasses import dataclass

import torch
import torch.nn.functional as F
from torch import nn

from .boxes import _pairwise_inter_union, box_cxcywh_to_xyxy, generalized_box_iou


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


def ota_cost(prob, boxes, target, center_radius=2.5, stride=32):
    """SimOTA cost of one image for a DETR-style query set (matcher.py:374-403, 405-446): class cost + 3 x (-GIoU), +100 where the query
    centre is not BOTH inside the target box and within `center_radius / stride` (normalised units) of its centre, +10000 on queries
    that are candidates of no target at all -> (cost (Q, T), pairwise IoU (Q, T))."""
    tb = target["boxes"]
    txy = box_cxcywh_to_xyxy(tb)
    qxy = box_cxcywh_to_xyxy(boxes)
    cx, cy = boxes[:, 0:1], boxes[:, 1:2]
    inside = (cx > txy[None, :, 0]) & (cx < txy[None, :, 2]) & (cy > txy[None, :, 1]) & (cy < txy[None, :, 3])
    r = center_radius / stride
    near = (cx > tb[None, :, 0] - r) & (cx < tb[None, :, 0] + r) & (cy > tb[None, :, 1] - r) & (cy < tb[None, :, 1] + r)
    candidate = inside.any(1) | near.any(1)
    inter, union = _pairwise_inter_union(qxy, txy)
    cost = class_cost(prob, target, "map") - 3.0 * generalized_box_iou(qxy, txy) + 100.0 * (~(inside & near)).to(prob.dtype)
    cost[~candidate] += 10000.0
    return cost, inter / union


def dynamic_k_assign(cost, iou):
    """SimOTA's dynamic-k assignment (matcher.py:448-509): target t takes its k_t cheapest queries, k_t = the (truncated, >= 1) sum of its 10
    largest IoUs; a query claimed by several targets keeps the cheapest; targets left empty then take their cheapest still-free query.
    `cost` is modified in place, as in the reference.  The reference evaluates "claimed by several targets" ONCE, before the repair loop, and
    re-uses that mask inside it -- kept, so that assignments agree case by case.
    -> ((query idx, target idx of each), the best query of every target)"""
    Q, T = cost.shape
    k = torch.clamp(iou.topk(min(Q, 10), dim=0)[0].sum(0).int(), min=1)
    match = torch.zeros_like(cost)
    for t in range(T):
        match[cost[:, t].topk(int(k[t]), largest=False)[1], t] = 1.0
    contested = match.sum(1) > 1

    def keep_cheapest():
        best = cost[contested].min(1)[1]
        match[contested] = 0
        match[contested, best] = 1

    if contested.any():
        keep_cheapest()
    while bool((match.sum(0) == 0).any()):
        cost[match.sum(1) > 0] += 100000.0
        for t in torch.nonzero(match.sum(0) == 0).flatten():
            match[cost[:, t].argmin(), t] = 1.0
        if bool((match.sum(1) > 1).any()):
            keep_cheapest()
    chosen = match.sum(1) > 0
    tgt = match[chosen].max(1)[1]
    cost[match == 0] = cost[match == 0] + float("inf")
    return (torch.nonzero(chosen).flatten(), tgt), cost.min(0)[1]


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
    def forward_ota(self, logits, boxes, targets):
        """`forward_ota` (matcher.py:347-372; MODEL.DDETRS.OTA, on in the shipped configs): one-to-many SimOTA matching of the decoder
        queries -> ([(query idx, target idx)] per image, [best query of every target] per image; images without targets: empty / [])."""
        pairs, best = [], []
        prob = logits.sigmoid()
        for b, t in enumerate(targets):
            if len(t["boxes"]) == 0:
                e = torch.zeros(0, dtype=torch.int64, device=logits.device)
                pairs.append((e, e.clone()))
                best.append([])
                continue
            p, q = dynamic_k_assign(*ota_cost(prob[b], boxes[b], t))
            pairs.append(p)
            best.append(q)
        return pairs, best

    @torch.no_grad()
    def forward_boxes_only(self, logits, boxes, targets):
        """the `force_box_loss` entry (matcher.py:640-729): class + L1 + GIoU, stuff targets keep their own box costs."""
        return [assign(cost_matrix(logits[b], boxes[b], t, self.w, stuff_takes_mean=False, class_mode=self.class_mode)) for b, t in enumerate(targets)]
