"""ONE TRAINING STEP of the path: the training branch of HIPIE_IMG.forward (hipie/hipie_img.py:287-313) around
DDETRSegmUniDN.coco_forward (hipie/models/ddetrs_dn.py:264-750) -- contrastive de-noising queries, the thing branch's transformer with
per-layer heads, the three groups of queries (foreground, background, de-noising) each with its own dynamic-mask predictions, matching and
DINO criterion call, the MaskDINO branch with its own de-noising part and criterion -- as a weighted loss dictionary over the PRODUCT
model's parameters, differentiable end to end (training/net.py).

    step = TrainStep(model)                                  # model: hipie_amd.hipie_img.HIPIE_IMG with loaded weights
    losses = step.loss_dict(batched_inputs, targets)          # dict of weighted scalars, as the reference's trainer sums them
    sum(losses.values()).backward()

tests/golden/train_step_tiny.npz holds the reference's own loss dictionary and every parameter gradient for one such step
(tests/golden/gen_train_step_golden.py); tests/test_training.py::test_train_step_* compare against it.

Three reference quirks are reproduced on purpose (they change the loss): the last decoder layer gets no box-IoU loss (post_processing
leaves 'pred_boxious' off the main dictionary), and the reference points handed to the dynamic mask head for the
de-noising and background queries are taken from the tensor that was ALREADY sliced to the foreground queries (ddetrs_dn.py:483 then
:491 / :531), in logit space and without the pixel scaling the foreground branch applies (:569-576)."""
import dataclasses

import torch
import torch.nn.functional as F

from . import net
from .criterion import DetCriterion, MaskCriterion
from .dn import cdn_queries, dn_match_indices, maskdino_dn_queries
from .matcher import HungarianMatcher, MatchWeights
from .targets import split_things_stuff
from .weights import maskdino_loss_plan, weighted_merge


class _Draws:
    """the step's uniform random numbers, in the reference's call order; default torch.rand"""

    def rand(self, shape, device):
        return torch.rand(shape, device=device)

    def randint(self, low, high, shape, device):
        return torch.randint(low, high, shape, device=device)


class TrainStep:
    """settings: configs/training/r50.yaml / vit_huge_32g.yaml (MODEL.DDETRS.*, MODEL.MASKDINO.LOSS_WEIGHT) and
    configs/mask_dino/maskdino_R50_bs16_50ep_3s_dowsample1_2048.yaml (the MaskDINO criterion)."""

    def __init__(self, model, backend=None, draws=None, dn_number=100, label_noise_ratio=0.5, box_noise_scale=1.0, fg_weight=1.0, bg_weight=0.0,
                 gt_weight=1.0, mask_dino_weight=1.0, num_points=112 * 112, bg_matcher_points=112 * 112, md_num_points=112 * 112, md_dn_number=100,
                 md_noise_scale=0.4, fusion_dropout=0.1, loss_weight=1.0,
                 weights=dict(loss_ce=2.0, loss_bbox=5.0, loss_giou=2.0, loss_mask=5.0, loss_dice=5.0)):
        self.model = model
        cfg = model.cfg
        self.cfg = dataclasses.asdict(cfg) if dataclasses.is_dataclass(cfg) else dict(cfg)
        self.be = backend or net.HipBackend
        self.draws = draws or _Draws()
        d = self.draws
        draw = lambda shape, device: d.rand(tuple(shape), device)          # noqa: E731
        self.dn_number, self.label_noise_ratio, self.box_noise_scale = dn_number, label_noise_ratio, box_noise_scale
        self.group_weights = (fg_weight, bg_weight, gt_weight)
        self.mask_dino_weight, self.fusion_dropout, self.loss_weight = mask_dino_weight, fusion_dropout, loss_weight
        self.md_dn_number, self.md_noise_scale = md_dn_number, md_noise_scale
        # hipie_img.py:176-236: the matchers and the DINO criterion
        self.matcher = HungarianMatcher(MatchWeights(2.0, 5.0, 2.0, 5.0, 5.0), num_points=num_points, stuff_takes_mean=True, draw=draw, class_mode="map")
        self.matcher_bg = HungarianMatcher(MatchWeights(2.0, 0.0, 0.0, 5.0, 5.0), num_points=bg_matcher_points, stuff_takes_mean=False, draw=draw, class_mode="map")
        self.criterion = DetCriterion(self.matcher, ["labelsVL", "boxes", "masks"], focal_alpha=0.25, mask_out_stride=self.cfg["mask_stride"],
                                      point_sample_masks=True, panoptic_box_loss=True, still_cls_for_encoder=True, num_points=num_points, draw=draw, ota=True)
        nd = self.cfg["dec_layers"]
        w = dict(weights)
        for i in range(nd - 1):
            w.update({k + "_%d" % i: v for k, v in weights.items()})
        w.update({k + "_enc": v for k, v in weights.items()})
        dnw = {"loss_ce_dn": weights["loss_ce"], "loss_bbox_dn": weights["loss_bbox"], "loss_giou_dn": weights["loss_giou"]}
        w.update(dnw)
        for i in range(nd - 1):
            w.update({k + "_%d" % i: v for k, v in dnw.items()})
        self.weight_dict = w
        # ddetrs_dn.py:176-196
        self.md_weights, md_dn_losses, md_matcher, md_losses = maskdino_loss_plan(
            4.0, 5.0, 5.0, 5.0, 2.0, True, "seg", True, self.cfg["md_dec_layers"], True, 4.0, 5.0, 5.0, 5.0, 2.0, md_num_points, True, draw=draw)
        self.md_criterion = MaskCriterion(100, md_matcher, md_losses, vl_loss=True, num_points=md_num_points, oversample_ratio=3.0, importance_sample_ratio=0.75,
                                          dn="seg", dn_losses=md_dn_losses, panoptic_on=False, draw=draw)

    # ---------------------------------------------------------------------------------------------------------------------------------
    def params(self):
        sd = dict(self.model.named_parameters(remove_duplicate=False))        # tied parameters (the shared heads) under every name they have
        sd.update(dict(self.model.named_buffers(remove_duplicate=False)))
        return sd

    def preprocess(self, images):
        """HIPIE_IMG.preprocess_image + nested_tensor_from_tensor_list(size_divisibility 32): list of (3,h,w) 0..255 -> batch, padding mask, sizes"""
        dev = self.model.pixel_mean.device
        mean, std = self.model.pixel_mean.view(3, 1, 1), self.model.pixel_std.view(3, 1, 1)
        sizes = [(int(x.shape[1]), int(x.shape[2])) for x in images]
        Hm, Wm = -(-max(s[0] for s in sizes) // 32) * 32, -(-max(s[1] for s in sizes) // 32) * 32
        t = torch.zeros(len(images), 3, Hm, Wm, device=dev)
        m = torch.ones(len(images), Hm, Wm, dtype=torch.bool, device=dev)
        for i, x in enumerate(images):
            t[i, :, :sizes[i][0], :sizes[i][1]] = (x.to(dev).float() - mean) / std
            m[i, :sizes[i][0], :sizes[i][1]] = False
        return t, m, sizes

    def mask_branch(self, memory, shapes, sd):
        """the CondInst mask features of forward_mask_head_train (ddetrs_dn.py:1006-1040): the first three encoder levels through MaskHeadSmallConv"""
        B, c = memory.shape[0], memory.shape[-1]
        lv, st = [], 0
        for (H, W) in shapes[:3]:
            lv.append(memory[:, st:st + H * W].reshape(B, H, W, c).permute(0, 3, 1, 2))
            st += H * W
        return net.mask_head_small_conv(lv, sd, "detr.mask_head.")

    def dyn_masks(self, mask_feats, ref_points, params, num_insts):
        """dynamic_mask_with_coords for per-image instance lists -> per image (1, n_i, 1, H/4, W/4) (forward_mask_head_train's output form)"""
        up = 8 // self.cfg["mask_stride"]
        m = self.be.dynamic_mask(mask_feats, ref_points, params, num_insts, 8, up)
        out, st = [], 0
        for n in num_insts:
            out.append(m[st:st + n][None, :, None])
            st += n
        return out

    # ---------------------------------------------------------------------------------------------------------------------------------
    def coco_forward(self, x, pad, sizes, targets, lang, task="detection"):
        """-> (unweighted loss dictionary as coco_forward returns it, debug outputs)"""
        sd, cfg, be, d = self.params(), self.cfg, self.be, self.draws
        dev = x.device
        gt_fg, gt_bg = split_things_stuff(targets)
        nbg, nq = cfg["num_bg_queries"], cfg["num_queries"]
        scale = torch.tensor([[float(w), float(h)] for (h, w) in sizes], device=dev)    # host -> device copies wait for the queue: made while it is empty
        # ---- contrastive de-noising queries; the label side is the image's (un-fused) text embedding (DYNAMIC_LABEL_ENC) (:324-360)
        pool0 = net.agg_lang_feat(lang["hidden"], lang["masks"])
        label_enc = net.ln(net.lin(pool0, sd, "detr.resizer.fc."), sd, "detr.resizer.layer_norm.", 1e-12)
        counts = [int(t["labels"].numel()) for t in targets]
        if max(len(t["labels"]) for t in gt_fg) == 0:
            raise NotImplementedError("training step: a batch without thing targets (the reference's no_fg branch) is not built")
        P, n_all = max(counts), sum(counts)
        G = max(1, self.dn_number // P)
        noise = None
        if self.box_noise_scale > 0:
            sign = d.randint(0, 2, (2 * G * n_all, 4), dev).float() * 2.0 - 1.0
            noise = {"sign": sign, "part": d.rand((2 * G * n_all, 4), dev)}
        q_label, q_box, attn_mask, dn_meta = cdn_queries(targets, self.dn_number, self.box_noise_scale, nq + nbg, label_enc, noise=noise,
                                                         label_noise_ratio=self.label_noise_ratio, num_classes=None)
        # The backbone is launched AFTER the de-noising queries, which need the text embedding and the targets only: cdn_queries ends in a
        # data-dependent torch.nonzero -- a host wait for everything queued so far.  Behind the backbone that wait is the backbone's whole GPU
        # time (114 ms per step of the host sitting there, sampled), and the host-bound matcher / criterion code that follows then runs with an
        # empty queue instead of behind it.
        feats, srcs, masks, poses = net.backbone_and_projections(x, pad, sd, cfg, be)
        tr = net.hipie_transformer(srcs, masks, poses, lang, sd, "detr.detr.transformer.", cfg, be, q_label, q_box, attn_mask, self.fusion_dropout)
        hs, memory, shapes = tr["hs"], tr["memory"], tr["shapes"]
        fused = {"hidden": tr["lang_hidden"], "masks": lang["masks"]}
        if task == "grounding":
            emb = net.agg_lang_feat(fused["hidden"], fused["masks"]).unsqueeze(1)
            text_masks = torch.ones(x.shape[0], 1, dtype=torch.bool, device=dev)
        else:
            emb, text_masks = fused["hidden"], fused["masks"]
        gt_indices = dn_match_indices(targets, dn_meta, dev)
        padding = dn_meta["single_padding"] * dn_meta["dn_num"]
        start_bg, start_fg = padding, padding + nbg
        mask_feats = self.mask_branch(memory, shapes, sd)
        B = x.shape[0]
        groups = {k: dict(cls=[], box=[], msk=[], idx=[]) for k in ("fg", "bg", "gt")}
        ious = []
        for lvl in range(hs.shape[0]):
            reference = net.inverse_sigmoid(tr["init_ref"] if lvl == 0 else tr["inter_refs"][lvl - 1])
            cls = net.vl_align(hs[lvl], emb, sd, "detr.detr.class_embed.%d." % lvl)
            box = (net.mlp(hs[lvl], sd, "detr.detr.bbox_embed.%d." % lvl, 3) + reference).sigmoid()
            iou = net.lin(hs[lvl], sd, "detr.detr.iou_head.%d." % lvl)
            params = net.mlp(hs[lvl], sd, "detr.controller.", 3)
            ref_fg = reference[:, start_fg:]                                            # (:483) -- everything below indexes THIS tensor
            # de-noising queries: fixed assignment (:487-521)
            g = groups["gt"]
            g["cls"].append(cls[:, :padding])
            g["box"].append(box[:, :padding])
            m = self.dyn_masks(mask_feats, ref_fg[:, :padding, :2].reshape(B * padding, 2), params[:, :padding].reshape(B * padding, -1), [padding] * B)
            g["msk"].append([mm[:, ids] for (ids, _), mm in zip(gt_indices, m)])
            g["idx"].append(gt_indices)
            # background queries: Mask2Former-style matching against the stuff targets (:523-551)
            g = groups["bg"]
            bcls, bbox = cls[:, start_bg:start_fg], box[:, start_bg:start_fg]
            m = self.dyn_masks(mask_feats, ref_fg[:, start_bg:start_fg, :2].reshape(B * nbg, 2), params[:, start_bg:start_fg].reshape(B * nbg, -1), [nbg] * B)
            bidx = self.matcher_bg(bcls, bbox, gt_bg, masks=[mm[0, :, 0] for mm in m], costs=("cls", "mask"))
            g["cls"].append(bcls)
            g["box"].append(bbox)
            g["msk"].append([mm[:, ids.to(dev)] for (ids, _), mm in zip(bidx, m)])
            g["idx"].append(bidx)
            # foreground queries: one-to-many SimOTA matching, masks of the matched queries only (:553-590)
            g = groups["fg"]
            fcls, fbox = cls[:, start_fg:], box[:, start_fg:]
            fidx, _ = self.matcher.forward_ota(fcls, fbox, gt_fg)
            pts = torch.cat([(ref_fg[i].sigmoid()[:, :2] * scale[i][None])[pi.to(dev)] for i, (pi, _) in enumerate(fidx)], 0)
            prm = torch.cat([params[i, start_fg:][pi.to(dev)] for i, (pi, _) in enumerate(fidx)], 0)
            g["cls"].append(fcls)
            g["box"].append(fbox)
            g["msk"].append(self.dyn_masks(mask_feats, pts, prm, [len(pi) for pi, _ in fidx]))
            g["idx"].append(fidx)
            ious.append(iou[:, start_fg:])

        # ---- MaskDINO branch with its own de-noising part and criterion (:630-668); its random draws come BEFORE the DINO criterion's
        md_losses = self.maskdino_losses(feats, sd, fused, emb, text_masks, targets)

        def assemble(g, with_iou=None, enc=False):                                     # post_processing (:769-796)
            o = {"pred_logits": g["cls"][-1], "pred_boxes": g["box"][-1], "pred_masks": g["msk"][-1], "text_masks": text_masks}
            o["aux_outputs"] = [{"pred_logits": a, "pred_boxes": b, "pred_masks": c, "text_masks": text_masks}
                                for a, b, c in zip(g["cls"][:-1], g["box"][:-1], g["msk"][:-1])]
            if with_iou is not None:               # the LAST layer's IoU predictions are not handed on (post_processing sets no 'pred_boxious'
                for a, u in zip(o["aux_outputs"], with_iou[:-1]):      # on the main dict, :779-789): loss_boxiou exists for the auxiliary layers only
                    a["pred_boxious"] = u
            if enc:
                o["enc_outputs"] = {"pred_logits": tr["enc_cls"], "pred_boxes": tr["enc_coord"].sigmoid(), "text_masks": text_masks}
            return o
        out_fg, out_bg, out_gt = assemble(groups["fg"], ious), assemble(groups["bg"]), assemble(groups["gt"], enc=True)
        l_fg = self.criterion(out_fg, gt_fg, groups["fg"]["idx"], dn_meta)
        l_bg = self.criterion(out_bg, gt_bg, groups["bg"]["idx"], dn_meta)
        l_gt = self.criterion(out_gt, targets, groups["gt"]["idx"], dn_meta)
        w = list(self.group_weights)
        losses = weighted_merge([l_fg, l_bg, l_gt], w)
        if task == "detection":
            losses.update(md_losses)
        return losses, dict(out_fg=out_fg, out_bg=out_bg, out_gt=out_gt)

    def maskdino_losses(self, feats, sd, fused, emb, text_masks, targets):
        cfg, be, d = self.cfg, self.be, self.draws
        dev = emb.device
        p = "detr.mask_dino.predictor."
        mf, ms = net.maskdino_pixel_decoder(feats, sd, "detr.mask_dino.pixel_decoder.", cfg, be)
        pool = net.agg_lang_feat(fused["hidden"], fused["masks"])
        label_enc = net.ln(net.lin(pool, sd, p + "resizer.fc."), sd, p + "resizer.layer_norm.", 1e-12)          # prepare_for_dn's dynamic label embedding
        counts = [int(t["labels"].numel()) for t in targets]
        P, n_all = max(counts), sum(counts)
        G = self.md_dn_number // P if P else 0
        dn, meta = None, None
        if G > 0:
            pr = d.rand((G * n_all,), dev)
            n_flip = int((pr < self.md_noise_scale * 0.5).sum())
            new = d.randint(0, cfg["hidden_dim"], (n_flip,), dev)                      # drawn and unused with the dynamic label embedding
            noise = {"p": pr, "new_label": new, "box": d.rand((G * n_all, 4), dev)}
            ql, qb, am, meta = maskdino_dn_queries(targets, self.md_dn_number, self.md_noise_scale, cfg["md_num_queries"], label_enc, noise=noise,
                                                   num_classes=cfg["hidden_dim"])
            dn = (ql, qb, am)
        r = net.maskdino_decoder(ms, mf, sd, p, cfg, be, dn=dn)
        L = cfg["md_dec_layers"]

        def vl(x, idx):
            return net.vl_align(x, emb, sd, "detr.mask_dino_cls_embed.%d." % idx)
        pad = meta["pad_size"] if meta else 0
        n_pred = len(r["classes"])                                                      # initial + L layers

        def part(lo, hi):
            o = {"pred_logits": vl(r["classes"][-1][:, lo:hi], L + 1), "pred_boxes": r["boxes"][-1][:, lo:hi], "pred_masks": r["masks"][-1][:, lo:hi],
                 "text_masks": text_masks}
            o["aux_outputs"] = [{"pred_logits": vl(r["classes"][i][:, lo:hi], i), "pred_boxes": r["boxes"][i][:, lo:hi], "pred_masks": r["masks"][i][:, lo:hi],
                                 "text_masks": text_masks} for i in range(n_pred - 1)]
            return o
        out = part(pad, None)
        it = r["interm"]
        out["interm_outputs"] = {"pred_logits": vl(it["pred_logits"], L), "pred_boxes": it["pred_boxes"], "pred_masks": it["pred_masks"], "text_masks": text_masks}
        mask_dict = None
        if meta:
            mask_dict = {"output_known_lbs_bboxes": part(0, pad), "scalar": meta["scalar"], "pad_size": pad}
        raw = self.md_criterion(out, targets, mask_dict)
        return {k + "_maskdino": v * self.md_weights[k] * self.mask_dino_weight for k, v in raw.items() if k in self.md_weights}

    # ---------------------------------------------------------------------------------------------------------------------------------
    def loss_dict(self, batched_inputs, targets, task="detection"):
        """batched_inputs: [{"image": (3,h,w) 0..255, "input_ids", "attention_mask"}]; targets: training.prepare_targets' dicts.
        -> the WEIGHTED loss dictionary of HIPIE_IMG.forward's training branch (hipie_img.py:301-312); sum its values for the total."""
        x, pad, sizes = self.preprocess([b["image"] for b in batched_inputs])
        with torch.no_grad():                                                           # MODEL.FREEZE_TEXT_ENCODER
            ids = torch.stack([b["input_ids"] for b in batched_inputs])
            am = torch.stack([b["attention_mask"] for b in batched_inputs])
            lang = self.model.text_encoder[0]({"input_ids": ids, "attention_mask": am}, sep=1012)
        lang = {"hidden": lang["hidden"].float(), "masks": lang["masks"]}
        targets = [{k: (v.to(x.device) if torch.is_tensor(v) else v) for k, v in t.items()} for t in targets]
        raw, _ = self.coco_forward(x, pad, sizes, targets, lang, task)
        out = {}
        for k, v in raw.items():
            if "_maskdino" in k:
                out[k] = v
            else:
                out[k] = v * (self.weight_dict.get(k, 1.0) * self.loss_weight if k in self.weight_dict else 1.0)
        return out


def build_optimizer(model, base_lr=1e-4, backbone_multiplier=0.1, weight_decay=0.01):
    """SOLVER of configs/training/*.yaml: AdamW, BASE_LR 1e-4, BACKBONE_MULTIPLIER 0.1, WEIGHT_DECAY 0.01 (the text encoder is frozen:
    MODEL.FREEZE_TEXT_ENCODER)."""
    bb, rest = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad or n.startswith("text_encoder."):
            continue
        (bb if ".backbone.0." in n else rest).append(p)
    # the multi-tensor (`fused`) implementation on the device: the same update in a handful of launches instead of several per parameter
    # (1241 tensors: 64 -> 13 ms per iteration at ViT-H)
    fused = bool(rest or bb) and all(p.is_cuda for p in rest + bb)
    return torch.optim.AdamW([{"params": rest, "lr": base_lr}, {"params": bb, "lr": base_lr * backbone_multiplier}], lr=base_lr,
                             weight_decay=weight_decay, fused=fused)


def train_iteration(step, optimizer, batched_inputs, targets, buckets=None, clip_norm=0.1, task="detection"):
    """one iteration of the reference's trainer (detectron2 SimpleTrainer.run_step under create_ddp_model, SOLVER.CLIP_GRADIENTS full_model
    0.1): forward -> summed loss -> backward (gradients all-reduced bucket by bucket while it runs: training/ddp.py) -> gradient clipping ->
    optimizer step.  Returns the loss dictionary (detached scalars) and the gradient norm before clipping."""
    optimizer.zero_grad(set_to_none=buckets is None)
    if buckets is not None:
        buckets.zero_grad()
    with torch.enable_grad():
        losses = step.loss_dict(batched_inputs, targets, task)
        total = sum(losses.values())
        total.backward()
    if buckets is not None:
        buckets.finish()
    params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
    norm = torch.nn.utils.clip_grad_norm_(params, clip_norm) if clip_norm else None
    optimizer.step()
    return {k: v.detach() for k, v in losses.items()}, norm
