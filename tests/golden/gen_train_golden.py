#!/usr/bin/env python3
"""Golden vectors of the reference's TRAINING-side host logic, made by running the reference's own classes on the CPU of the
build container:  tests/golden/train_matcher.npz, train_dn.npz, train_criterion.npz, train_maskdino.npz.

    python tests/golden/gen_train_golden.py          (needs /root/reference; never runs on the GPU box)

What is imported from /root/reference and run as is: HungarianMatcherVL (models/deformable_detr/matcher.py), MaskDINO's HungarianMatcher
(models/maskdino/matcher.py), SetCriterion / DINOCriterion (models/deformable_detr/deformable_detr.py), MaskDINO's SetCriterion
(models/maskdino/criterion.py), DDETRSegmUniDN.prepare_for_cdn / compute_gt_indices / dn_post_process (models/ddetrs_dn.py) and
PointRend's point_sample / get_uncertain_point_coords_with_randomness (projects/PointRend/point_rend/point_features.py, vendored).
What the container lacks and is supplied here: fvcore.nn.giou_loss (fvcore is not vendored: the published per-pair formula; the test
additionally checks it against 1 - diag of the reference's own pairwise generalized_box_iou) and a CUDA device (`.cuda()` / `.to("cuda")`
are made no-ops for the duration of a call).  Every random tensor the reference draws (torch.rand / rand_like / randint_like) is RECORDED
in call order and stored in the fixture: the build's functions take them as inputs, so the comparison is exact, not statistical."""
import contextlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()
M_DET = ref_shim.ref("models.deformable_detr.matcher")
M_MD = ref_shim.ref("models.maskdino.matcher")
C_DET = ref_shim.ref("models.deformable_detr.deformable_detr")
C_MD = ref_shim.ref("models.maskdino.criterion")
DN = ref_shim.ref("models.ddetrs_dn")
BOX = ref_shim.ref("util.box_ops")

_spec = importlib.util.spec_from_file_location("ref_point_features", os.path.join(ref_shim.REF_ROOT, "projects/PointRend/point_rend/point_features.py"))
PF = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(PF)
PF.cat = lambda ts, dim=0: torch.cat(list(ts), dim)          # detectron2.layers.cat (a thin torch.cat wrapper) is a placeholder in the shim


def fvcore_giou_loss(boxes1, boxes2, reduction="none", eps=1e-7):
    """fvcore.nn.giou_loss as published (fvcore/nn/giou_loss.py): per pair, xyxy; reduction none | mean | sum"""
    x1, y1, x2, y2 = boxes1.unbind(-1)
    x1g, y1g, x2g, y2g = boxes2.unbind(-1)
    xk1, yk1, xk2, yk2 = torch.max(x1, x1g), torch.max(y1, y1g), torch.min(x2, x2g), torch.min(y2, y2g)
    inter = torch.zeros_like(x1)
    m = (yk2 > yk1) & (xk2 > xk1)
    inter[m] = (xk2[m] - xk1[m]) * (yk2[m] - yk1[m])
    union = (x2 - x1) * (y2 - y1) + (x2g - x1g) * (y2g - y1g) - inter
    iou = inter / (union + eps)
    area_c = (torch.max(x2, x2g) - torch.min(x1, x1g)) * (torch.max(y2, y2g) - torch.min(y1, y1g))
    loss = 1 - (iou - (area_c - union) / (area_c + eps))
    return loss.mean() if reduction == "mean" else loss.sum() if reduction == "sum" else loss


for mod in (M_DET, M_MD, C_DET, C_MD):
    mod.point_sample = PF.point_sample
    if hasattr(mod, "get_uncertain_point_coords_with_randomness"):
        mod.get_uncertain_point_coords_with_randomness = PF.get_uncertain_point_coords_with_randomness
C_DET.giou_loss = fvcore_giou_loss
# TorchScript's `num_boxes: int` rejects the float count the criterion computes (clamp(...).item()): the eager twin of the same function
C_DET.sigmoid_focal_loss_jit = C_DET.sigmoid_focal_loss
C_MD.get_world_size = lambda: 1                # detectron2.utils.comm is a placeholder in the shim: one process
sys.modules["torchvision"]._is_tracing = lambda: False      # the shim's torchvision placeholder answers every call with a truthy object


@contextlib.contextmanager
def cpu_as_cuda(record):
    """inside: .cuda() / .to("cuda") stay on the CPU; every torch.rand / rand_like / randint_like result is appended to `record`"""
    keep = (torch.Tensor.cuda, torch.Tensor.to, torch.rand, torch.rand_like, torch.randint_like)
    t_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x.startswith("cuda")) else x for x in a)
        k = {kk: ("cpu" if (isinstance(v, str) and v.startswith("cuda")) else v) for kk, v in k.items()}
        return t_to(self, *a, **k)

    def rec(fn):
        def w(*a, **k):
            k = {kk: ("cpu" if (isinstance(v, str) and v.startswith("cuda")) else v) for kk, v in k.items()}
            r = fn(*a, **k)
            record.append(r.clone())
            return r
        return w
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.to = to
    torch.rand, torch.rand_like, torch.randint_like = rec(keep[2]), rec(keep[3]), rec(keep[4])
    try:
        yield
    finally:
        torch.Tensor.cuda, torch.Tensor.to, torch.rand, torch.rand_like, torch.randint_like = keep


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%s: %d arrays, %.1f KB" % (name, len(out), os.path.getsize(path) / 1024))


def rand_boxes(g, n):
    c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    s = torch.rand(n, 2, generator=g) * 0.3 + 0.05
    return torch.cat((c, s), 1)


def make_targets(g, counts, L, hw, stuff):
    ts = []
    for i, n in enumerate(counts):
        pm = torch.zeros(n, L, dtype=torch.bool)
        for t in range(n):
            a = int(torch.randint(1, L - 3, (1,), generator=g))
            pm[t, a:a + int(torch.randint(1, 3, (1,), generator=g))] = True
        thing = torch.ones(n, dtype=torch.bool)
        for s_ in stuff.get(i, ()):
            thing[s_] = False
        ts.append({"labels": torch.randint(0, 7, (n,), generator=g), "boxes": rand_boxes(g, n), "positive_map": pm, "is_thing": thing,
                   "masks": (torch.rand(n, hw[i][0], hw[i][1], generator=g) > 0.6).float()})
    return ts


def flat(prefix, targets):
    d = {}
    for i, t in enumerate(targets):
        for k, v in t.items():
            d["%s%d_%s" % (prefix, i, k)] = v
    return d


def pairs(prefix, idx):
    d = {}
    for i, (a, b) in enumerate(idx):
        d["%s%d_q" % (prefix, i)], d["%s%d_t" % (prefix, i)] = a, b
    return d


def gen_matcher():
    g = torch.Generator().manual_seed(11)
    B, Q, L = 3, 24, 16
    targets = make_targets(g, (4, 3, 2), L, [(32, 40)] * 3, {0: (1,), 2: (0, 1)})       # image 2: stuff only (the NaN -> 0 path)
    logits = torch.randn(B, Q, L, generator=g)
    boxes = torch.stack([rand_boxes(g, Q) for _ in range(B)])
    pmasks = [torch.randn(1, Q, 1, 8, 10, generator=g) * 3 for _ in range(B)]
    m = M_DET.HungarianMatcherVL(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, cost_mask=5.0, cost_dice=5.0, panoptic_box_loss=True)
    rec = []
    with cpu_as_cuda(rec):
        i_box = m.memory_efficient_forward({"pred_logits": logits, "pred_boxes": boxes}, targets)
        i_mask = m.memory_efficient_forward({"pred_logits": logits, "pred_boxes": boxes, "pred_masks": pmasks}, targets)
        i_enc = m.forward({"pred_logits": logits[..., :1], "pred_boxes": boxes},
                          [dict(t, positive_map=torch.ones(len(t["boxes"]), 1, dtype=torch.bool)) for t in targets], force_box_loss=True)
        cls0 = m.compute_cost_label_vl(0, logits.sigmoid(), targets)
    # cost pieces of image 0 with the recorded points (for a tolerance check of the terms themselves)
    pts = rec[0]
    om = PF.point_sample(pmasks[0][0, :, 0][:, None], pts.repeat(Q, 1, 1), align_corners=False).squeeze(1)
    tm = PF.point_sample(targets[0]["masks"][:, None], pts.repeat(4, 1, 1), align_corners=False).squeeze(1)
    arrays = dict(logits=logits, boxes=boxes, pmasks=torch.stack([p[0, :, 0] for p in pmasks]), cls0=cls0,
                  ce0=M_DET.batch_sigmoid_ce_loss(om, tm), dice0=M_DET.batch_dice_loss(om, tm),
                  giou0=BOX.generalized_box_iou(BOX.box_cxcywh_to_xyxy(boxes[0]), BOX.box_cxcywh_to_xyxy(targets[0]["boxes"])),
                  weights=np.array([2.0, 5.0, 2.0, 5.0, 5.0]))
    arrays.update(flat("t", targets)); arrays.update(pairs("box", i_box)); arrays.update(pairs("mask", i_mask)); arrays.update(pairs("enc", i_enc))
    for i, r in enumerate(rec):
        arrays["rand%d" % i] = r
    # MaskDINO's matcher: class ids, its own point count, box costs on, panoptic
    md = M_MD.HungarianMatcher(cost_class=4.0, cost_mask=5.0, cost_dice=5.0, num_points=300, cost_box=5.0, cost_giou=2.0, panoptic_on=True, vl_loss=False)
    lg7 = torch.randn(B, Q, 7, generator=g)
    pm3 = torch.randn(B, Q, 8, 10, generator=g) * 3
    rec2 = []
    with cpu_as_cuda(rec2):
        i_md = md.memory_efficient_forward({"pred_logits": lg7, "pred_boxes": boxes, "pred_masks": pm3}, targets)
        md.vl_loss = True
        i_md_vl = md.memory_efficient_forward({"pred_logits": logits, "pred_boxes": boxes, "pred_masks": pm3}, targets)
    arrays.update(md_logits=lg7, md_masks=pm3, md_weights=np.array([4.0, 5.0, 2.0, 5.0, 5.0]))
    arrays.update(pairs("md", i_md)); arrays.update(pairs("mdvl", i_md_vl))
    for i, r in enumerate(rec2):
        arrays["md_rand%d" % i] = r
    # SimOTA (the shipped configs train with MODEL.DDETRS.OTA on).  torchvision.ops.box_iou is a placeholder in the shim: the reference's own
    # pairwise IoU (util/box_ops.py:34-46) stands in for it -- the same published formula
    M_DET.ops.box_iou = lambda a, b: BOX.box_iou(a, b)[0]
    ota_boxes = boxes.clone()
    for b_, t in enumerate(targets):                      # put a few queries on the targets so that IoUs (and dynamic k) are not all tiny
        for j in range(len(t["boxes"])):
            ota_boxes[b_, 3 * j:3 * j + 3] = t["boxes"][j] + 0.02 * torch.randn(3, 4, generator=g)
    ota_boxes[..., 2:] = ota_boxes[..., 2:].clamp(min=0.02)
    empty = {"labels": torch.zeros(0, dtype=torch.long), "boxes": torch.zeros(0, 4), "positive_map": torch.zeros(0, L, dtype=torch.bool),
             "is_thing": torch.zeros(0, dtype=torch.bool), "masks": torch.zeros(0, 32, 40)}
    with cpu_as_cuda([]):
        i_ota, best = m.forward_ota({"pred_logits": logits, "pred_boxes": ota_boxes}, targets[:2] + [empty])
    arrays["ota_boxes"] = ota_boxes
    # a contended case: three targets of which two coincide, four queries -- the repair loop of dynamic_k_matching has to run
    tb = torch.tensor([[0.5, 0.5, 0.3, 0.3], [0.5, 0.5, 0.3, 0.3], [0.2, 0.2, 0.1, 0.1]])
    tie_t = [{"labels": torch.zeros(3, dtype=torch.long), "boxes": tb, "positive_map": targets[0]["positive_map"][:3], "is_thing": torch.ones(3, dtype=torch.bool)}]
    tie_q = torch.tensor([[[0.5, 0.5, 0.3, 0.3], [0.52, 0.5, 0.3, 0.28], [0.8, 0.8, 0.1, 0.1], [0.21, 0.2, 0.1, 0.1]]])
    tie_l = torch.randn(1, 4, L, generator=g)
    with cpu_as_cuda([]):
        i_tie, best_tie = m.forward_ota({"pred_logits": tie_l, "pred_boxes": tie_q}, tie_t)
    arrays.update(tie_boxes=tb, tie_queries=tie_q, tie_logits=tie_l, tie_best=best_tie[0])
    arrays.update(pairs("tie", i_tie))
    arrays.update(pairs("ota", i_ota))
    for i, q in enumerate(best):
        arrays["ota_best%d" % i] = q if torch.is_tensor(q) else torch.zeros(0, dtype=torch.long)
    save("train_matcher", **arrays)


class _Self:
    pass


def gen_dn():
    g = torch.Generator().manual_seed(12)
    arrays = {}
    for tag, dynamic, counts, dn_number, ratio in (("dyn", True, (3, 5, 0), 20, 0.5), ("ids", False, (2, 1), 7, 0.5), ("one", True, (4,), 2, 0.0)):
        targets = [{"labels": torch.randint(0, 6, (n,), generator=g), "boxes": rand_boxes(g, n)} for n in counts]
        me = _Self()
        me.dynamic_label_enc = dynamic
        C = 8
        table = torch.randn(6, C, generator=g)
        emb = torch.randn(len(counts), C, generator=g) if dynamic else (lambda ids: table[ids])
        rec = []
        with cpu_as_cuda(rec):
            ql, qb, mask, meta = DN.DDETRSegmUniDN.prepare_for_cdn(me, targets, dn_number, ratio, 0.4 if tag != "one" else 1.0, 10, 6, C, emb)
            idx = DN.DDETRSegmUniDN.compute_gt_indices(me, targets, meta["dn_num"], meta["dp_num"], meta["single_padding"])
        arrays.update({tag + "_label": ql, tag + "_box": qb, tag + "_mask": mask, tag + "_meta": np.array([meta["single_padding"], meta["dn_num"], meta["dp_num"]]),
                       tag + "_emb": emb if dynamic else table, tag + "_args": np.array([dn_number, ratio, 0.4 if tag != "one" else 1.0, 10, 6])})
        arrays.update(flat(tag + "_t", targets)); arrays.update(pairs(tag + "_idx", idx))
        for i, r in enumerate(rec):
            arrays["%s_rand%d" % (tag, i)] = r
    # MaskDINO's own (DN-DETR style) de-noising queries: maskdino_decoder.py:202-327
    MDD = ref_shim.ref("models.maskdino.transformer_decoder.maskdino_decoder")
    for tag, dynamic, counts, dn_num in (("md_dyn", True, (3, 2), 7), ("md_ids", False, (2, 4, 0), 9)):
        targets = [{"labels": torch.randint(0, 6, (n,), generator=g), "boxes": rand_boxes(g, n)} for n in counts]
        C, Qn = 8, 5
        me = _Self()
        me.training, me.dn_num, me.noise_scale, me.num_classes, me.hidden_dim, me.num_queries, me.dynamic_label_enc = True, dn_num, 0.4, 6, C, Qn, dynamic
        table = torch.randn(6, C, generator=g)
        pooled = torch.randn(len(counts), C, generator=g)
        me.resizer = lambda x: x
        me.label_enc = lambda ids: table[ids]
        tgt, refp = torch.randn(Qn, C, generator=g), torch.randn(Qn, 4, generator=g)
        rec = []
        with cpu_as_cuda(rec):
            ql, qb, mask, md = MDD.MaskDINODecoder.prepare_for_dn(me, targets, tgt, refp, len(counts), pooled if dynamic else None)
        arrays.update({tag + "_label": ql, tag + "_box": qb, tag + "_mask": mask, tag + "_meta": np.array([md["pad_size"], md["scalar"]]),
                       tag + "_emb": pooled if dynamic else table, tag + "_args": np.array([dn_num, 0.4, Qn, 6]), tag + "_tgt": tgt, tag + "_refp": refp})
        arrays.update(flat(tag + "_t", targets))
        for i, r in enumerate(rec):
            arrays["%s_rand%d" % (tag, i)] = r
    save("train_dn", **arrays)


def _scalars(prefix, d):
    return {prefix + k: (v if torch.is_tensor(v) else torch.as_tensor(v)).detach().float().reshape(()) for k, v in d.items()}


def gen_criterion():
    g = torch.Generator().manual_seed(13)
    B, Q, L, layers = 2, 20, 12, 3
    targets = make_targets(g, (3, 4), L, [(64, 96), (64, 64)], {1: (2,)})
    text_masks = torch.ones(B, L, dtype=torch.long)
    text_masks[:, -3:] = 0

    def layer_out():
        return {"pred_logits": torch.randn(B, Q, L, generator=g), "pred_boxes": torch.stack([rand_boxes(g, Q) for _ in range(B)]),
                "pred_boxious": torch.randn(B, Q, 1, generator=g), "text_masks": text_masks}
    outs = [layer_out() for _ in range(layers)]
    matcher = M_DET.HungarianMatcherVL(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, cost_mask=5.0, cost_dice=5.0)
    rec = []
    with cpu_as_cuda(rec):
        indices = [matcher.memory_efficient_forward({"pred_logits": o["pred_logits"], "pred_boxes": o["pred_boxes"]}, targets) for o in outs]
    # matched-instance masks of the last layer in assignment order: per image (1, n_i, 1, H/4, W/4) with H, W the /32-padded batch size (64 x 96)
    for li, o in enumerate(outs):
        o["pred_masks"] = [torch.randn(1, len(indices[li][b][0]), 1, 16, 24, generator=g) * 2 for b in range(B)]
    pred_masks = outs[-1]["pred_masks"]
    outputs = dict(outs[-1])
    outputs["aux_outputs"] = [dict(o) for o in outs[:-1]]
    outputs["enc_outputs"] = {"pred_logits": torch.randn(B, Q, 1, generator=g), "pred_boxes": torch.stack([rand_boxes(g, Q) for _ in range(B)]), "text_masks": text_masks}
    # de-noising part: 2 groups, single_padding = 2 * 4
    G, SP = 2, 8
    known = {"pred_logits": torch.randn(B, G * SP, L, generator=g), "pred_boxes": torch.stack([rand_boxes(g, G * SP) for _ in range(B)]), "text_masks": text_masks}
    known["aux_outputs"] = [{"pred_logits": torch.randn(B, G * SP, L, generator=g), "pred_boxes": torch.stack([rand_boxes(g, G * SP) for _ in range(B)]),
                             "text_masks": text_masks} for _ in range(layers - 1)]
    dn_meta = {"single_padding": SP, "dn_num": G, "dp_num": 0, "output_known_lbs_bboxes": known}
    crit = C_DET.DINOCriterion(matcher, {}, ["labelsVL", "boxes", "masks"], focal_alpha=0.25, mask_out_stride=4, still_cls_for_encoder=True,
                               point_sample=True, panoptic_box_loss=True)
    crit.num_points = 500
    rec2 = []
    with cpu_as_cuda(rec2):
        losses = crit(outputs, targets, indices, dn_metas=dn_meta)
        crit.point_sample = False
        dense = crit.loss_masks(outputs, targets, indices[-1], 7.0)
        nodn = crit.compute_dn_loss(None, targets, layers - 1, 7.0)
    # OTA: one-to-many pairs, every loss per matched pair
    M_DET.ops.box_iou = lambda a, b: BOX.box_iou(a, b)[0]
    ota_out = {"pred_logits": outs[-1]["pred_logits"], "pred_boxes": outs[-1]["pred_boxes"].clone(), "pred_boxious": outs[-1]["pred_boxious"], "text_masks": text_masks}
    for b_, t in enumerate(targets):
        for j in range(len(t["boxes"])):
            ota_out["pred_boxes"][b_, 2 * j:2 * j + 2] = (t["boxes"][j] + 0.02 * torch.randn(2, 4, generator=g)).clamp(min=0.02)
    rec3 = []
    with cpu_as_cuda(rec3):
        ota_idx, _ = matcher.forward_ota(ota_out, targets)
        ota_out["pred_masks"] = [torch.randn(1, len(ota_idx[b_][0]), 1, 16, 24, generator=g) * 2 for b_ in range(B)]
        crit_ota = C_DET.DINOCriterion(matcher, {}, ["labelsVL", "boxes", "masks"], focal_alpha=0.25, mask_out_stride=4, ota=True, point_sample=True)
        crit_ota.num_points = 300
        ota_losses = C_DET.SetCriterion.forward(crit_ota, ota_out, targets, [ota_idx])
    arrays = dict(text_masks=text_masks, enc_logits=outputs["enc_outputs"]["pred_logits"], enc_boxes=outputs["enc_outputs"]["pred_boxes"],
                  meta=np.array([SP, G, layers]), known_logits=known["pred_logits"], known_boxes=known["pred_boxes"], giou_pairs_ref=torch.zeros(1))
    for i, o in enumerate(outs):
        arrays.update({"l%d_logits" % i: o["pred_logits"], "l%d_boxes" % i: o["pred_boxes"], "l%d_boxious" % i: o["pred_boxious"]})
        arrays.update(pairs("l%d_idx" % i, indices[i]))
    for i, a in enumerate(known["aux_outputs"]):
        arrays.update({"known_aux%d_logits" % i: a["pred_logits"], "known_aux%d_boxes" % i: a["pred_boxes"]})
    for li, o in enumerate(outs):
        for b in range(B):
            arrays["l%d_pred_masks%d" % (li, b)] = o["pred_masks"][b]
    arrays.update(flat("t", targets))
    arrays.update(_scalars("loss_", losses)); arrays.update(_scalars("dense_", dense)); arrays.update(_scalars("nodn_", nodn))
    for i, r in enumerate(rec2):
        arrays["rand%d" % i] = r
    arrays.update(ota_boxes=ota_out["pred_boxes"], ota_masks0=ota_out["pred_masks"][0], ota_masks1=ota_out["pred_masks"][1])
    arrays.update(pairs("ota_idx", ota_idx)); arrays.update(_scalars("otaloss_", ota_losses))
    for i, r in enumerate(rec3):
        arrays["ota_rand%d" % i] = r
    # the per-pair GIoU loss against the reference's own pairwise GIoU (the only third-party formula of the criterion)
    a, b = BOX.box_cxcywh_to_xyxy(rand_boxes(g, 40)), BOX.box_cxcywh_to_xyxy(rand_boxes(g, 40))
    arrays.update(giou_a=a, giou_b=b, giou_pairwise_diag=torch.diag(BOX.generalized_box_iou(a, b)), iou_diag=C_DET.compute_box_iou(a, b))
    save("train_criterion", **arrays)


def gen_maskdino():
    g = torch.Generator().manual_seed(14)
    B, Q, NC, L = 2, 18, 7, 12
    targets = make_targets(g, (3, 2), L, [(40, 48), (32, 48)], {0: (0,)})
    for t in targets:
        t["labels"] = torch.randint(0, NC, t["labels"].shape, generator=g)

    def out(nq=Q, ncls=NC):
        return {"pred_logits": torch.randn(B, nq, ncls, generator=g), "pred_boxes": torch.stack([rand_boxes(g, nq) for _ in range(B)]),
                "pred_masks": torch.randn(B, nq, 10, 12, generator=g) * 2}
    outputs = out()
    outputs["aux_outputs"] = [out(), out()]
    outputs["interm_outputs"] = out()
    groups, single = 2, 3
    known = out(groups * single)
    known["aux_outputs"] = [out(groups * single), out(groups * single)]
    mask_dict = {"output_known_lbs_bboxes": known, "known_indice": torch.zeros(5), "scalar": groups, "pad_size": groups * single}
    matcher = M_MD.HungarianMatcher(cost_class=4.0, cost_mask=5.0, cost_dice=5.0, num_points=200, cost_box=5.0, cost_giou=2.0, panoptic_on=True, vl_loss=False)
    crit = C_MD.SetCriterion(NC, matcher, {}, 0.1, ["labels", "masks", "boxes"], False, 200, 3.0, 0.75, dn="seg", dn_losses=["labels", "masks", "boxes"],
                             panoptic_on=True)
    rec = []
    with cpu_as_cuda(rec):
        losses = crit(outputs, targets, mask_dict)
    rec2 = []
    with cpu_as_cuda(rec2):
        nodn = crit({k: v for k, v in outputs.items() if k != "interm_outputs"}, targets, None)
    arrays = dict(meta=np.array([groups, single, NC]))

    def put(prefix, o):
        for k in ("pred_logits", "pred_boxes", "pred_masks"):
            arrays[prefix + k] = o[k]
    put("main_", outputs); put("aux0_", outputs["aux_outputs"][0]); put("aux1_", outputs["aux_outputs"][1]); put("interm_", outputs["interm_outputs"])
    put("known_", known); put("known_aux0_", known["aux_outputs"][0]); put("known_aux1_", known["aux_outputs"][1])
    arrays.update(flat("t", targets)); arrays.update(_scalars("loss_", losses)); arrays.update(_scalars("nodn_", nodn))
    for i, r in enumerate(rec):
        arrays["rand%d" % i] = r
    for i, r in enumerate(rec2):
        arrays["nodn_rand%d" % i] = r
    save("train_maskdino", **arrays)


def gen_targets():
    hipie_img_mod = ref_shim.ref_hipie_img()          # hipie_img.py with detectron2's vendored (dependency-free) Boxes / Instances
    st = {"instances": types.SimpleNamespace(Instances=hipie_img_mod.Instances), "boxes": types.SimpleNamespace(Boxes=hipie_img_mod.Boxes)}
    g = torch.Generator().manual_seed(15)
    arrays = {}
    insts = []
    for i, (n, hw) in enumerate(((3, (48, 64)), (0, (32, 32)), (2, (40, 24)))):
        inst = st["instances"].Instances(hw)
        xy0 = torch.rand(n, 2, generator=g) * torch.tensor([hw[1], hw[0]]) * 0.5
        wh = torch.rand(n, 2, generator=g) * torch.tensor([hw[1], hw[0]]) * 0.4 + 1
        inst.gt_boxes = st["boxes"].Boxes(torch.cat((xy0, xy0 + wh), 1))
        inst.gt_classes = torch.randint(0, 9, (n,), generator=g)
        inst.positive_map = torch.rand(n, 6, generator=g) > 0.6
        inst.is_thing = torch.rand(n, generator=g) > 0.4
        inst.gt_masks = (torch.rand(n, hw[0], hw[1], generator=g) > 0.5)
        insts.append(inst)
        arrays.update({"in%d_boxes" % i: inst.gt_boxes.tensor, "in%d_classes" % i: inst.gt_classes, "in%d_pm" % i: inst.positive_map,
                       "in%d_thing" % i: inst.is_thing, "in%d_masks" % i: inst.gt_masks, "in%d_hw" % i: np.array(hw)})
    me = _Self()
    me.device, me.use_amp, me.use_lsj = torch.device("cpu"), False, True
    out = hipie_img_mod.HIPIE_IMG.prepare_targets(me, insts)
    for i, t in enumerate(out):
        for k, v in t.items():
            arrays["out%d_%s" % (i, k)] = v
    save("train_targets", **arrays)


def gen_weights():
    import json
    out = {}
    for tag, kw in (("seg", dict(TWO_STAGE=True, DN="seg", DEEP_SUPERVISION=True, BOX_LOSS=True)), ("std", dict(TWO_STAGE=False, DN="standard", DEEP_SUPERVISION=True, BOX_LOSS=False)),
                    ("no", dict(TWO_STAGE=True, DN="no", DEEP_SUPERVISION=False, BOX_LOSS=True))):
        md = types.SimpleNamespace(CLASS_WEIGHT=4.0, COST_CLASS_WEIGHT=4.0, COST_DICE_WEIGHT=5.0, DICE_WEIGHT=5.0, COST_MASK_WEIGHT=5.0, NO_OBJECT_WEIGHT=0.1,
                                   MASK_WEIGHT=5.0, COST_BOX_WEIGHT=5.0, BOX_WEIGHT=5.0, COST_GIOU_WEIGHT=2.0, GIOU_WEIGHT=2.0, DEC_LAYERS=3, TRAIN_NUM_POINTS=12544, **kw)
        cfg = types.SimpleNamespace(MODEL=types.SimpleNamespace(MaskDINO=md))
        w, dn_losses, matcher, losses = DN.get_weight_dict(cfg, True)
        out[tag] = {"weights": w, "dn_losses": dn_losses, "losses": losses, "args": kw,
                    "matcher": [matcher.cost_class, matcher.cost_box, matcher.cost_giou, matcher.cost_mask, matcher.cost_dice, matcher.num_points, bool(matcher.vl_loss),
                                bool(matcher.panoptic_on)]}
    with open(os.path.join(HERE, "train_weights.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("train_weights.json: %d plans" % len(out))


if __name__ == "__main__":
    which = sys.argv[1:] or ["matcher", "dn", "criterion", "maskdino", "weights", "targets"]
    for w in which:
        globals()["gen_" + w]()
