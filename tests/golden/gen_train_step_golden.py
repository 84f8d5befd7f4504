#!/usr/bin/env python3
"""Golden vector of ONE TRAINING STEP of the reference: DDETRSegmUniDN.coco_forward (models/ddetrs_dn.py:264-750) with the
reference's own criteria (DINOCriterion x 3 calls: foreground / background / de-noising queries, MaskDINO's SetCriterion) on the CPU of the
build container, followed by loss.backward() -- tests/golden/train_step_tiny.npz holds the loss dictionary, the weighted total and the
gradient of every trainable parameter (large ones as strided subsamples).

    python tests/golden/gen_train_step_golden.py          (needs /root/reference; never runs on the GPU box)

Model = gen_golden.build_ref_model(TINY) (the e2e_tiny configuration: the reference's classes assembled as hipie_img.py:77-176 does), weights
and images from _synth as in e2e_tiny, training settings from configs/training/r50.yaml (OTA, POINT_SAMPLE, DYNAMIC_LABEL_ENC, DN_NUMBER 100,
FINAL_BG_WEIGHT 0, FINAL_GT_WEIGHT 1).  What the container lacks is supplied as in gen_train_golden.py (fvcore's giou_loss formula, PointRend's
point_sample from the vendored project, `.cuda()` as a no-op); every random tensor the step draws comes from _synth.HashDraws (a counter-based hash, identical on every device) in call order, so the
product's step can draw the same numbers without storing them."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _synth  # noqa: E402
import gen_golden as GG  # noqa: E402
import gen_train_golden as GT  # noqa: E402

ref = GG.ref


def build(c):
    model = GG.build_ref_model(c)
    bert = GG.build_ref_bert(c)
    man = _synth.load_synth(model, seed=71, prefix="detr.")
    man_b = _synth.load_synth(bert, seed=72, prefix="text_encoder.body.")
    # training attributes of DDETRSegmUniDN.__init__ (ddetrs_dn.py:144-215) with configs/training/r50.yaml
    model.debug_only, model.aux_loss, model.enc_mask = False, True, False
    model.num_queries, model.background_proposals, model.embed_dim = c["num_queries"], c["num_bg_queries"], 256
    model.dynamic_label_enc, model.bg_query_from_lang = True, False
    # DN_NUMBER 12 instead of 100: coco_forward takes the reference points of the de-noising queries from the FOREGROUND-sliced tensor
    # (ddetrs_dn.py:483,491), which needs padding_size + background queries <= the number of foreground queries -- 900 at the shipped size, 40 here
    model.dn_number, model.dp_number, model.label_noise_ratio, model.box_noise_scale = 12, 0, 0.5, 1.0
    model.bg_weight, model.fg_weight, model.gt_weight = 0.0, 1.0, 1.0
    model.background_matcher, model.mask_dino_weight = "Mask2Former", 1.0
    model.still_cls_for_encoder = True
    return model, bert, man, man_b


def criteria(c):
    M_DET, C_DET, M_MD, C_MD = GT.M_DET, GT.C_DET, GT.M_MD, GT.C_MD
    M_DET.ops.box_iou = lambda a, b: GT.BOX.box_iou(a, b)[0]          # torchvision.ops.box_iou (torchvision is a placeholder in the shim): the reference's own pairwise IoU
    matcher = M_DET.HungarianMatcherVL(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, cost_mask=5.0, cost_dice=5.0, panoptic_box_loss=True)
    MM = ref("models.deformable_detr.matcher_mask")             # hipie_img.py:16: HungarianMatcher as HungarianMatcherBG
    MM.point_sample = GT.PF.point_sample
    matcher_bg = MM.HungarianMatcher(2.0, 5.0, 5.0)
    crit = C_DET.DINOCriterion(matcher, {}, ["labelsVL", "boxes", "masks"], focal_alpha=0.25, ota=True, still_cls_for_encoder=True,
                               point_sample=True, matcher_bg=matcher_bg, panoptic_box_loss=True)
    return crit


def maskdino_criterion(dec_layers, num_points):
    """DDETRSegmUniDN.__init__ (ddetrs_dn.py:176-196) + get_weight_dict (:36-87) with configs/mask_dino/maskdino_R50_bs16_50ep_3s_dowsample1_2048.yaml:
    CLASS 4 / MASK 5 / DICE 5 / BOX 5 / GIOU 2 and the same costs, TWO_STAGE, DN "seg", deep supervision, PANO_BOX_LOSS off, vl_loss (the
    class head is the vision-language one: FIXED_LINEAR_HEAD off)."""
    M_MD, C_MD = GT.M_MD, GT.C_MD
    weight = {"loss_ce": 4.0, "loss_mask": 5.0, "loss_dice": 5.0, "loss_bbox": 5.0, "loss_giou": 2.0}
    weight.update({k + "_interm": v for k, v in list(weight.items())})
    weight.update({k + "_dn": v for k, v in list(weight.items())})
    weight.update({k + "_%d" % i: v for i in range(dec_layers) for k, v in list(weight.items())})
    matcher = M_MD.HungarianMatcher(cost_class=4.0, cost_mask=5.0, cost_dice=5.0, cost_box=5.0, cost_giou=2.0, num_points=num_points, vl_loss=True)
    return C_MD.SetCriterion(100, matcher=matcher, weight_dict=weight, eos_coef=0.1, losses=["labels", "masks", "boxes"], vl_loss=True,
                             num_points=num_points, oversample_ratio=3.0, importance_sample_ratio=0.75, dn="seg",
                             dn_losses=["labels", "masks", "boxes"], panoptic_on=False, semantic_ce_loss=False)


def synth_targets(sizes, L, seed=77):
    """two images' ground truth in prepare_targets' format (hipie_img.py:422-447): masks at the padded batch size, boxes cxcywh / image size"""
    g = torch.Generator().manual_seed(seed)
    Hm, Wm = max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes)
    Hm, Wm = -(-Hm // 32) * 32, -(-Wm // 32) * 32
    ts = GT.make_targets(g, (3, 4), L, [(Hm, Wm)] * len(sizes), {1: (2,)})
    for t, (h, w) in zip(ts, sizes):
        t["image_size"] = torch.tensor([w, h, w, h], dtype=torch.float)
        t["masks"][:, h:, :] = 0
        t["masks"][:, :, w:] = 0
    return ts


import contextlib


@contextlib.contextmanager
def hashed_randomness(draws):
    """inside: .cuda() / .to("cuda") stay on the CPU (GT.cpu_as_cuda) and torch.rand / rand_like / randint_like return _synth.HashDraws values
    (call order = the reference's)"""
    with GT.cpu_as_cuda([]):
        keep = (torch.rand, torch.rand_like, torch.randint_like)

        def rand(*size, **k):
            size = size[0] if len(size) == 1 and not isinstance(size[0], int) else size
            return draws.rand(tuple(size)).to(k.get("dtype") or torch.float32)

        def rand_like(t, **k):
            return draws.rand(tuple(t.shape)).to(t.dtype if t.is_floating_point() else torch.float32)

        def randint_like(t, *a, **k):
            if "high" in k:
                low, high = k.get("low", 0), k["high"]
            else:
                low, high = (0, a[0]) if len(a) == 1 else (a[0], a[1])
            return draws.randint(int(low), int(high), tuple(t.shape)).to(k.get("dtype") or t.dtype)
        torch.rand, torch.rand_like, torch.randint_like = rand, rand_like, randint_like
        try:
            yield
        finally:
            torch.rand, torch.rand_like, torch.randint_like = keep


def main():
    c = dict(GG.TINY)
    model, bert, man, man_b = build(c)
    sizes = [(200, 256), (256, 224)]
    imgs = _synth.synth_images(sizes, seed=73)
    mean = torch.tensor(c["pixel_mean"]).view(3, 1, 1)
    std = torch.tensor(c["pixel_std"]).view(3, 1, 1)
    Hm, Wm = max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes)
    batched = torch.zeros(len(imgs), 3, Hm, Wm)
    for i, x in enumerate(imgs):
        batched[i, :, :x.shape[1], :x.shape[2]] = (x - mean) / std
    images = GG._ImageList(batched, sizes)
    ids, mask, pmap = _synth.synth_token_ids(2, 9, 64, seed=74)
    L = int(ids.shape[1])
    targets = synth_targets(sizes, L)
    crit = criteria(c)
    crit.num_points = 400                             # TRAIN_NUM_POINTS scaled to the 64 x 64 mask grid of this fixture (12544 = 112^2 at the shipped size)
    model.mask_dino_criterion = maskdino_criterion(c["md_dec_layers"], 400)
    print("criterion ok")
    model.train()                                     # MaskDINODecoder keys its de-noising branch and the per-layer mask predictions on self.training
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0                                 # all 0.0 already (DROPOUT 0.0) except MaskDINO's DYNAMIC_LABEL_ENC_DROPOUT 0.1: off, so the step is deterministic
        if isinstance(getattr(m, "dropout", None), float):
            m.dropout = 0.0                           # BiMultiHeadAttention's functional attention dropout (fuse_helper.py:111-112, 0.1 in vlfusion.py:81): off likewise
    draws = _synth.HashDraws()
    torch.set_grad_enabled(True)
    for p in bert.parameters():
        p.requires_grad_(False)                     # MODEL.FREEZE_TEXT_ENCODER
    with hashed_randomness(draws):
        lang = bert({"input_ids": ids, "attention_mask": mask}, sep=1012)
        out, loss_dict = model.coco_forward(images, targets, crit, train=True,
                                            language_dict_features={"hidden": lang["hidden"].clone(), "masks": lang["masks"]}, task="detection")
    # HIPIE_IMG.forward's training branch (hipie_img.py:301-312): criterion weights x SOLVER.LOSS_WEIGHT_DET (1.0); the MaskDINO entries carry
    # their weights already; entries without a weight (loss_boxiou_*) stay as they are -- the trainer sums every entry of the dictionary
    W = {"loss_ce": 2.0, "loss_bbox": 5.0, "loss_giou": 2.0, "loss_mask": 5.0, "loss_dice": 5.0}
    wd = dict(W)
    for i in range(c["dec_layers"] - 1):
        wd.update({k + "_%d" % i: v for k, v in W.items()})
    wd.update({k + "_enc": v for k, v in W.items()})
    Wdn = {"loss_ce_dn": 2.0, "loss_bbox_dn": 5.0, "loss_giou_dn": 2.0}
    wd.update(Wdn)
    for i in range(c["dec_layers"] - 1):
        wd.update({k + "_%d" % i: v for k, v in Wdn.items()})
    raw = {k: v.detach().clone() for k, v in loss_dict.items()}
    total = 0.0
    for k in loss_dict:
        w = 1.0 if "_maskdino" in k else wd.get(k, 1.0)
        total = total + loss_dict[k] * w
    for p_ in model.parameters():
        p_.grad = None
    total.backward()
    arrays = {"total": total.detach(), "n_rand": np.array(draws.calls)}
    arrays.update({"loss/" + k: v for k, v in raw.items()})
    arrays.update({"weight/" + k: np.array(1.0 if "_maskdino" in k else wd.get(k, 1.0)) for k in raw})
    arrays.update({k: (v.to(torch.uint8) if k.endswith("_masks") else v) for k, v in GT.flat("t", targets).items()})
    meta_sub = {}
    n_with, n_zero = 0, 0
    for name, p_ in model.named_parameters():
        g = torch.zeros_like(p_) if p_.grad is None else p_.grad
        n_with += int(p_.grad is not None)
        n_zero += int(p_.grad is None)
        g = g.detach().reshape(-1)
        if g.numel() > 1024:
            step = -(-g.numel() // 1024)
            meta_sub["detr." + name] = step
            g = g[::step].clone()
        arrays["grad/detr." + name] = g
        arrays["gnorm/detr." + name] = (torch.zeros(()) if p_.grad is None else p_.grad.detach().double().norm().float())
    arrays["grad_steps"] = np.frombuffer(__import__("json").dumps(meta_sub).encode(), dtype=np.uint8)
    # intermediates for stage-wise debugging of a re-implementation
    for tag in ("out_fg", "out_bg", "out_gt"):
        o = out[tag]
        arrays[tag + "/pred_logits"] = o["pred_logits"].detach()
        arrays[tag + "/pred_boxes"] = o["pred_boxes"].detach()
    arrays["lang_hidden"] = lang["hidden"].detach()
    arrays["cfg_json"] = np.frombuffer(__import__("json").dumps(dict(cfg=c, sizes=sizes, dn_number=12, num_points=400, n_classes=9, max_len=64,
                                                                    manifest={**{"detr." + k: list(v) for k, v in man.items()},
                                                                              **{"text_encoder.body." + k: list(v) for k, v in man_b.items()}})).encode(), dtype=np.uint8)
    GT.save("train_step_tiny", **arrays)
    print("total %.6f; %d parameters with a gradient, %d without; %d random draws" % (float(total), n_with, n_zero, draws.calls))
    print({k: round(float(v), 4) for k, v in list(raw.items())[:12]})




if __name__ == "__main__":
    main()
