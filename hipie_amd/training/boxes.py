"""Box arithmetic of the losses and matching costs (hipie/util/box_ops.py:17-90; fvcore.nn.giou_loss; deformable_detr.py:922-938)."""
import torch


def box_cxcywh_to_xyxy(b):
    """(cx, cy, w, h) -> (x0, y0, x1, y1)   (util/box_ops.py:17-23)"""
    c, s = b[..., :2], b[..., 2:]
    return torch.cat((c - 0.5 * s, c + 0.5 * s), -1)


def _area(b):
    return (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])


def _pairwise_inter_union(a, b):
    lo = torch.maximum(a[:, None, :2], b[None, :, :2])
    hi = torch.minimum(a[:, None, 2:], b[None, :, 2:])
    wh = (hi - lo).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return inter, _area(a)[:, None] + _area(b)[None, :] - inter


def generalized_box_iou(a, b):
    """pairwise GIoU of xyxy boxes, (N, 4) x (M, 4) -> (N, M); the enclosing-box term carries the reference's 1e-7
    (util/box_ops.py:64-86).  Degenerate boxes are a caller error there (assert) and here (ValueError)."""
    if not bool((a[:, 2:] >= a[:, :2]).all() & (b[:, 2:] >= b[:, :2]).all()):          # one host wait for both checks
        raise ValueError("generalized_box_iou: boxes with x1 < x0 or y1 < y0")
    inter, union = _pairwise_inter_union(a, b)
    lo = torch.minimum(a[:, None, :2], b[None, :, :2])
    hi = torch.maximum(a[:, None, 2:], b[None, :, 2:])
    wh = (hi - lo).clamp(min=0)
    hull = wh[..., 0] * wh[..., 1]
    return inter / union - (hull - union) / (hull + 1e-7)


def paired_iou(a, b):
    """IoU of box i of `a` with box i of `b` (xyxy): the diagonal deformable_detr.py:922-938 takes of the pairwise matrix."""
    lo = torch.maximum(a[:, :2], b[:, :2])
    hi = torch.minimum(a[:, 2:], b[:, 2:])
    wh = (hi - lo).clamp(min=0)
    inter = wh[:, 0] * wh[:, 1]
    return inter / (_area(a) + _area(b) - inter)


def paired_giou_loss(a, b, eps=1e-7):
    """1 - GIoU of box i of `a` with box i of `b` (xyxy), per pair.  The reference calls fvcore.nn.giou_loss(reduction="none")
    (deformable_detr.py:438; fvcore is not vendored: this is its published formula -- intersection only where the boxes overlap,
    eps in both quotients)."""
    x1 = torch.maximum(a[:, 0], b[:, 0])
    y1 = torch.maximum(a[:, 1], b[:, 1])
    x2 = torch.minimum(a[:, 2], b[:, 2])
    y2 = torch.minimum(a[:, 3], b[:, 3])
    overlap = (y2 > y1) & (x2 > x1)
    inter = torch.where(overlap, (x2 - x1) * (y2 - y1), torch.zeros_like(x1))
    union = _area(a) + _area(b) - inter
    iou = inter / (union + eps)
    hull = (torch.maximum(a[:, 2], b[:, 2]) - torch.minimum(a[:, 0], b[:, 0])) * (torch.maximum(a[:, 3], b[:, 3]) - torch.minimum(a[:, 1], b[:, 1]))
    return 1 - (iou - (hull - union) / (hull + eps))
