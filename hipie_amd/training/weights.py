"""Loss weighting of the training forward (hipie/models/ddetrs_dn.py:36-88 ``get_weight_dict``, :233-242 ``merge_dict``)."""
from .matcher import HungarianMatcher, MatchWeights


def maskdino_loss_plan(class_weight, mask_weight, dice_weight, box_weight, giou_weight, two_stage, dn, deep_supervision, dec_layers, box_loss,
                       cost_class, cost_mask, cost_dice, cost_box, cost_giou, train_num_points, vl_loss, draw=None):
    """MODEL.MaskDINO.* -> (weight of every loss key the MaskDINO criterion can emit, the loss families of the de-noising part, the matcher,
    the loss families).  Key growth as in the reference: base keys, their `_interm` copies (two-stage), `_dn` copies of all of those (dn ==
    "seg") or of the non-mask ones ("standard"), then `_i` copies of everything for each of the `dec_layers` auxiliary layers."""
    w = {"loss_ce": class_weight, "loss_mask": mask_weight, "loss_dice": dice_weight, "loss_bbox": box_weight, "loss_giou": giou_weight}
    if two_stage:
        w.update({k + "_interm": v for k, v in list(w.items())})
    if dn == "standard":
        w.update({k + "_dn": v for k, v in list(w.items()) if k not in ("loss_mask", "loss_dice")})
        dn_losses = ["labels", "boxes"]
    elif dn == "seg":
        w.update({k + "_dn": v for k, v in list(w.items())})
        dn_losses = ["labels", "masks", "boxes"]
    else:
        dn_losses = []
    if deep_supervision:
        base = list(w.items())
        for i in range(dec_layers):
            w.update({k + "_%d" % i: v for k, v in base})
    matcher = HungarianMatcher(MatchWeights(cost_class, cost_box, cost_giou, cost_mask, cost_dice), num_points=train_num_points, stuff_takes_mean=False,
                               draw=draw, class_mode="map" if vl_loss else "ids")
    return w, dn_losses, matcher, (["labels", "masks", "boxes"] if box_loss else ["labels", "masks"])


def weighted_merge(dicts, weights):
    """sum_k w_k * d_k[key] over the loss dicts that hold `key`, in order (merge_dict)."""
    out = {}
    for d, w in zip(dicts, weights):
        for k, v in d.items():
            out[k] = v * w if k not in out else out[k] + v * w
    return out
