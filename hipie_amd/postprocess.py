"""Per-image post-processing after the parity surface (HIPIE_IMG.inference, hipie_img.py:537-766).

Round-1 scope: token logits -> class scores (convert_grounding_to_od_logits, hipie_img.py:1025-1052, mean pooling over each
class's token span), score = sqrt(sigmoid(cls) * sigmoid(iou)) (hipie_img.py:610-617), top-k instances with their masks
up-sampled to the image and thresholded.  NMS, CLIP fusion and the panoptic merge are the next row.
"""
import torch
import torch.nn.functional as F


_PMAP_CACHE = {}


def positive_map_matrix(positive_map, L, device):
    """(L, num_classes) matrix M with M[t, c] = 1/len(tokens of class c+1) so that token scores @ M is the per-class mean of
    convert_grounding_to_od_logits (hipie_img.py:1041-1049) -- one GEMM for the whole batch instead of a Python loop of
    per-class gathers.  Cached per prompt (the prompt is fixed over an evaluation run)."""
    key = (id(positive_map), L, str(device))
    m = _PMAP_CACHE.get(key)
    if m is None:
        m = torch.zeros(L, len(positive_map))
        for label_j, toks in positive_map.items():
            m[list(toks), int(label_j) - 1] = 1.0 / len(toks)
        m = m.to(device)
        _PMAP_CACHE[key] = m
    return m


def convert_grounding_to_od_logits(logits, num_classes, positive_map):
    """logits (..., Q, L) token scores -> (..., Q, num_classes) mean over each class's token span."""
    return logits @ positive_map_matrix(positive_map, logits.shape[-1], logits.device)


def inference(model, out, batched_inputs, topk=100):
    """batched over the images (no per-image / per-class Python loops on the device path)."""
    task = batched_inputs[0]["task"]
    nbg = model.cfg.num_bg_queries
    sizes = out["image_sizes"]
    B = len(batched_inputs)
    logits = out["pred_logits"][:, nbg:].float().sigmoid()               # (B, Q, L)
    iou = out["pred_boxious"][:, nbg:].float().sigmoid()                 # (B, Q, 1)
    if task == "grounding":
        cls = logits
    else:
        pmap = batched_inputs[0].get("positive_map_label_to_token", {1: [0]})
        cls = convert_grounding_to_od_logits(logits, len(pmap), pmap)
    score = torch.sqrt(cls * iou)                                         # (B, Q, C)
    C = score.shape[-1]
    k = min(topk, score.shape[1] * C)
    top, idx = score.flatten(1).topk(k, dim=1)                            # (B, k)
    qi, ci = idx // C, idx % C
    boxes = torch.gather(out["pred_boxes"][:, nbg:].float(), 1, qi.unsqueeze(-1).expand(-1, -1, 4))
    wh = torch.tensor([[w, h, w, h] for (h, w) in sizes], dtype=torch.float32, device=boxes.device).unsqueeze(1) \
        if not hasattr(model, "_wh_cache") or model._wh_cache[0] != tuple(sizes) else model._wh_cache[1]
    model._wh_cache = (tuple(sizes), wh)
    cx, cy, bw, bh = boxes.unbind(-1)
    xyxy = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], -1) * wh
    pm = out["pred_masks"][:, nbg:, 0]                                    # (B, Q, H/4, W/4)
    m = torch.gather(pm, 1, qi[:, :, None, None].expand(-1, -1, pm.shape[-2], pm.shape[-1]))
    m = F.interpolate(m.float(), scale_factor=model.mask_stride, mode="bilinear", align_corners=False) > 0.0   # sigmoid > 0.5
    results = []
    for i in range(B):
        h, w = sizes[i]
        results.append({"instances": {"pred_boxes": xyxy[i], "scores": top[i], "pred_classes": ci[i],
                                      "pred_masks": m[i, :, :h, :w]}})
    return results
