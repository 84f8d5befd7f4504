"""Per-image post-processing after the parity surface (HIPIE_IMG.inference, hipie_img.py:537-766).

Round-1 scope: token logits -> class scores (convert_grounding_to_od_logits, hipie_img.py:1025-1052, mean pooling over each
class's token span), score = sqrt(sigmoid(cls) * sigmoid(iou)) (hipie_img.py:610-617), top-k instances with their masks
up-sampled to the image and thresholded.  NMS, CLIP fusion and the panoptic merge are the next row.
"""
import torch
import torch.nn.functional as F


def convert_grounding_to_od_logits(logits, num_classes, positive_map):
    """logits (Q, L) token logits -> (Q, num_classes) by averaging each class's token positions (hipie_img.py:1041-1049)."""
    scores = torch.zeros(logits.shape[0], num_classes, device=logits.device)
    for label_j, toks in positive_map.items():
        scores[:, int(label_j) - 1] = logits[:, torch.as_tensor(toks, device=logits.device)].mean(-1)
    return scores


def inference(model, out, batched_inputs, topk=100):
    results = []
    task = batched_inputs[0]["task"]
    nbg = model.cfg.num_bg_queries
    for i, inp in enumerate(batched_inputs):
        h, w = out["image_sizes"][i]
        logits = out["pred_logits"][i][nbg:].sigmoid()
        iou = out["pred_boxious"][i][nbg:].sigmoid()
        if task == "grounding":
            cls = logits
        else:
            pmap = inp.get("positive_map_label_to_token", {1: [0]})
            cls = convert_grounding_to_od_logits(logits, len(pmap), pmap)
        score = torch.sqrt(cls * iou)
        k = min(topk, score.numel())
        top, idx = score.flatten().topk(k)
        qi, ci = idx // score.shape[1], idx % score.shape[1]
        boxes = out["pred_boxes"][i][nbg:][qi]
        cx, cy, bw, bh = boxes.unbind(-1)
        xyxy = torch.stack([(cx - bw / 2) * w, (cy - bh / 2) * h, (cx + bw / 2) * w, (cy + bh / 2) * h], -1)
        m = out["pred_masks"][i][nbg:][qi]                               # (k,1,H/4,W/4)
        m = F.interpolate(m, scale_factor=model.mask_stride, mode="bilinear", align_corners=False)[:, 0, :h, :w]
        results.append({"instances": {"pred_boxes": xyxy, "scores": top, "pred_classes": ci, "pred_masks": m.sigmoid() > 0.5}})
    return results
