"""Post-processing after the parity surface, on the device and batched.

Mirrors HIPIE_IMG.inference (projects/HIPIE/hipie/hipie_img.py:537-766), panoptic_inference (:473-535), semantic_inference
(:870-878), convert_grounding_to_od_logits (:1025-1052) and segmentation_postprocess (hipie/models/ddetrs.py:1029-1076)
for decouple_decoder True / bg_query_from_lang False / demo_only False / score_thres 0 (the shipped eval settings), with the
MaskCLIP score fusion of MODEL.CLIP.ENABLED (hipie_img.py:592-609, 735-747, 811-868; hipie_amd/open_vocab.py) when the model has it.

What changed relative to the reference's per-image, per-class, per-segment Python loops:
  * token -> class pooling is one GEMM (mean) or one padded gather + max for the whole batch, FG/BG masking a `where`;
  * NMS is hipie_batched_nms (one launch for the batch, suppression matrix in LDS, bit-exact keep lists);
  * the top-100 selection, box conversion / scaling / clipping are batched; one host sync fetches the per-image counts;
  * instance masks go through hipie_mask_finalize (x4 bilinear -> sigmoid -> threshold -> crop -> nearest resize, one pass);
  * the panoptic merge has no .item() loop: segment areas are bincounts, the first-come-first-served id assignment with
    stuff merging is a cumulative sum + first-occurrence scatter, the label map one LUT gather; segments_info is built
    from a single small device->host copy per batch.
"""
import torch
import torch.nn.functional as F

from . import ops
from .structures import Boxes, Instances

NEG = -9999.0                                              # the reference's "class not allowed" logit (hipie_img.py:1047-1050)
_CACHE = {}


def _cached(key, build):
    v = _CACHE.get(key)
    if v is None:
        if len(_CACHE) > 64:
            _CACHE.clear()
        v = _CACHE[key] = build()
    return v


def _pmap_key(positive_map):
    return tuple((int(k), tuple(v)) for k, v in positive_map.items())


def positive_map_matrix(positive_map, L, device):
    """(L, C) matrix M, M[t, c] = 1/len(tokens of class c+1): token logits @ M is the per-class mean of
    convert_grounding_to_od_logits (hipie_img.py:1041-1046).  Cached per prompt."""
    def build():
        m = torch.zeros(L, len(positive_map))
        for label_j, toks in positive_map.items():
            m[list(toks), int(label_j) - 1] = 1.0 / len(toks)
        return m.to(device)
    return _cached(("mean", _pmap_key(positive_map), L, str(device)), build)


def positive_map_table(positive_map, device):
    """(C, T) padded token-index table + validity mask for the max-pool variant (TEST.MAX_POOL, hipie_img.py:1043-1044)."""
    def build():
        C = len(positive_map)
        T = max(len(v) for v in positive_map.values())
        idx = torch.zeros(C, T, dtype=torch.long)
        ok = torch.zeros(C, T, dtype=torch.bool)
        for label_j, toks in positive_map.items():
            idx[int(label_j) - 1, :len(toks)] = torch.as_tensor(list(toks))
            ok[int(label_j) - 1, :len(toks)] = True
        return idx.to(device), ok.to(device)
    return _cached(("max", _pmap_key(positive_map), str(device)), build)


def thing_vector(is_thing, num_classes, device):
    """(C) bool, True where class c+1 is a thing (missing keys default to thing, hipie_img.py:509,1047)."""
    key = ("thing", tuple(sorted((int(k), bool(v)) for k, v in is_thing.items())), num_classes, str(device))
    return _cached(key, lambda: torch.tensor([bool(is_thing.get(c + 1, True)) for c in range(num_classes)], device=device))


def convert_grounding_to_od_logits(logits, num_classes, positive_map, is_thing=None, mode=None, model_free=False, max_pool=False):
    """logits (B, Q, L) -> (B, Q, num_classes).  is_thing: list (one dict per image) or a single dict; mode: None | "FG" | "BG"
    or a list of them per image."""
    assert logits.dim() == 3
    B = logits.shape[0]
    if max_pool:
        idx, ok = positive_map_table(positive_map, logits.device)
        scores = torch.where(ok, logits[:, :, idx], logits.new_tensor(float("-inf"))).amax(-1)
    else:
        scores = logits @ positive_map_matrix(positive_map, logits.shape[-1], logits.device)
    modes = list(mode) if isinstance(mode, (list, tuple)) else [mode] * B
    if model_free or all(m is None for m in modes):
        return scores
    things = is_thing if isinstance(is_thing, (list, tuple)) else [is_thing or {}] * B
    block = []
    for b in range(B):
        tv = thing_vector(things[b], num_classes, logits.device)
        block.append(~tv if modes[b] == "FG" else tv if modes[b] == "BG" else torch.zeros_like(tv))
    return torch.where(torch.stack(block)[:, None, :], scores.new_tensor(NEG), scores)


def _cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1)


# ------------------------------------------------------------------------------------------------ panoptic / semantic
def _sem_pan(cls_all, masks_lo, stride, crop_hw, out_hw, thing_vec, cfg, precision=0):
    """semantic_inference + the tensor part of panoptic_inference for one image.
    cls_all (N, C) class probabilities, masks_lo (N, hm, wm) stride-`stride` logits.  Returns sem_seg (C, oh, ow) and a
    dict of device tensors describing the panoptic result (label map + per-segment table).
    On the device the N x H x W sigmoid tensor is never built (hipie_sem_pan); the torch formulation below it is the same
    arithmetic for CPU tensors (host-logic tests) and for shapes the kernel does not cover."""
    N, C = cls_all.shape
    scores, labels = cls_all.max(-1)
    kept = scores > cfg.object_mask_threshold
    if cls_all.is_cuda and ops.sem_pan_ok(N, C):
        sem, ids, own, orig_area = ops.sem_pan(masks_lo.float().contiguous(), cls_all.float().contiguous(),
                                               torch.where(kept, scores, scores.new_tensor(-1.0)), stride, crop_hw, out_hw,
                                               precision)
        ids = ids.long()
        some = ids >= 0
        own = own & some
        ids = ids.clamp_min(0)
        flat = ids.view(-1)
        mask_area = torch.bincount(flat[some.view(-1)], minlength=N)
    else:
        up = F.interpolate(masks_lo[:, None].float(), scale_factor=float(stride), mode="bilinear", align_corners=False)
        up = up[:, :, :crop_hw[0], :crop_hw[1]]
        if tuple(out_hw) != tuple(crop_hw):
            up = F.interpolate(up, size=tuple(out_hw), mode="bilinear", align_corners=False)
        sig = up[:, 0].contiguous().sigmoid_()                                      # (N, oh, ow)
        oh, ow = sig.shape[-2:]
        sem = (cls_all.t() @ sig.view(N, -1)).view(C, oh, ow)                       # einsum("qc,qhw->chw")
        w = torch.where(kept, scores, scores.new_tensor(-1.0))                      # never wins the argmax unless nothing is kept
        ids = (w.view(-1, 1, 1) * sig).argmax(0)                                    # (oh, ow) index into N
        own = sig.gather(0, ids[None])[0] >= 0.5                                    # the winner's own mask >= 0.5
        flat = ids.view(-1)
        mask_area = torch.bincount(flat, minlength=N)
        orig_area = (sig >= 0.5).view(N, -1).sum(1)
    inter_area = torch.bincount(flat[own.view(-1)], minlength=N)
    ratio_ok = mask_area.double() / orig_area.clamp_min(1).double() >= cfg.overlap_threshold
    valid = kept & (mask_area > 0) & (orig_area > 0) & (inter_area > 0) & ratio_ok
    isthing = thing_vec[labels]
    # first-come-first-served ids; a stuff class re-uses the id of its first valid segment (hipie_img.py:519-527)
    ar = torch.arange(N, device=cls_all.device)
    stuff = valid & ~isthing
    first_of_class = torch.full((C,), N, dtype=torch.long, device=cls_all.device)
    first_of_class.scatter_reduce_(0, labels[stuff], ar[stuff], reduce="amin")
    is_first = stuff & (first_of_class[labels] == ar)
    new = valid & (isthing | is_first)
    new_id = torch.cumsum(new.to(torch.int32), 0, dtype=torch.int32)
    seg_id = torch.where(new, new_id, new_id[first_of_class[labels].clamp_max(N - 1)])
    seg_id = torch.where(valid, seg_id, torch.zeros_like(seg_id))
    pan = torch.where(own, seg_id[ids], torch.zeros_like(seg_id[ids])).to(torch.int32)
    return sem, dict(pan=pan, new=new, labels=labels, isthing=isthing, valid=valid)


def _segments_info(tables):
    """one device->host copy for the whole batch -> the reference's segments_info lists (hipie_img.py:529-535)."""
    if not tables:
        return []
    lens = [int(t["new"].numel()) for t in tables]
    packed = torch.cat([torch.stack([t["new"].long(), t["labels"].long(), t["isthing"].long()]) for t in tables], 1).cpu()
    out, st = [], 0
    for n in lens:
        new, lab, thing = packed[:, st:st + n].tolist()
        st += n
        info, sid = [], 0
        for k in range(n):
            if new[k]:
                sid += 1
                info.append({"id": sid, "isthing": bool(thing[k]), "category_id": int(lab[k])})
        out.append(info)
    return out


def _clip_vocab(model, batched_inputs):
    test = batched_inputs[0].get("open_seg_labels")
    if test is None:
        raise ValueError("MODEL.CLIP.ENABLED needs `open_seg_labels` in every input (the test-time mapper provides it)")
    if getattr(model, "train_labels", None) is None:
        raise RuntimeError("MaskCLIP: the training vocabulary is missing -- set model.train_labels (get_openseg_labels('coco_panoptic', "
                           "prompt_engineered=True)) or provide the reference's openseg_labels directory (HIPIE_ASSETS)")
    return [t["name"].split(",") for t in test], [t["name"].split(",") for t in model.train_labels]


def _clip_states(model, batched_inputs, out):
    """the mask-independent pass of MaskCLIP's visual tower over the images of this call (open_vocab.MaskCLIP.encode_images), ONCE per
    inference call: both fusion sites (instances, semantic / panoptic) read it.  Images of one size go through the tower as one batch.
    Kept in the output dict of the call, so its lifetime is the call's.  The image is the un-normalised input / 255 (hipie_img.py:349-351)."""
    st = out.get("_clip_states")
    if st is None:
        dev = out["pred_logits"].device
        imgs = [x["image"].to(dev).float() / 255.0 for x in batched_inputs]
        if len(set(tuple(t.shape) for t in imgs)) == 1:
            st = ("batched", model.clip.encode_images(torch.stack(imgs)))
        else:
            st = ("list", [model.clip.encode_images(t[None]) for t in imgs])
        out["_clip_states"] = st
    return st


def _clip_logits_all(model, batched_inputs, out, mask_logits, pred_open_prob, blocked=None):
    """HIPIE_IMG.get_clip_logits (hipie_img.py:811-868) for every image of the call: mask_logits / pred_open_prob = per-image lists of
    (Q_i, h, w) / (Q_i, C) -> list of fused (Q_i, C).  The test vocabulary is the inputs' `open_seg_labels` ([{"name": "a,b,..."}] per
    class, data/coco_dataset_mapper_uni.py; one vocabulary per call, as one task per call: hipie_img.py:285), the training vocabulary
    model.train_labels."""
    from .open_vocab import get_clip_logits, get_clip_logits_batched
    test, train = _clip_vocab(model, batched_inputs)
    cfg = model.cfg
    kind, st = _clip_states(model, batched_inputs, out)
    if kind == "batched" and all(x.get("open_seg_labels") == batched_inputs[0].get("open_seg_labels") for x in batched_inputs):
        return get_clip_logits_batched(model.clip, st, mask_logits, test, train, pred_open_prob, cfg.clip_alpha, cfg.clip_beta, cfg.clip_agg_mode,
                                       blocked=blocked)
    res = []
    for i, x in enumerate(batched_inputs):
        test_i = [t["name"].split(",") for t in x["open_seg_labels"]]
        state = st[i] if kind == "list" else model.clip.state_of(st, i)
        res.append(get_clip_logits(model.clip, None, mask_logits[i], test_i, train, pred_open_prob[i], cfg.clip_alpha, cfg.clip_beta,
                                   cfg.clip_agg_mode, state=state))
    return res


# ------------------------------------------------------------------------------------------------ the entry point
@torch.no_grad()
def _instances_on_device(model, out, batched_inputs, do_postprocess=True):
    """the batched, device-side part of HIPIE_IMG.inference (hipie_img.py:600-668 + segmentation_postprocess): class scores,
    NMS, the per-image top-k over (kept query, class), boxes scaled / clipped to the output size.  No host synchronisation:
    every per-image quantity is a row of a (B, K) tensor and `ok` marks the rows that are instances."""
    cfg = model.cfg
    use_clip = bool(getattr(model, "enable_clip", False))
    task = batched_inputs[0]["task"]
    max_num_inst = {"detection": 100, "grounding": 1}[task]
    nbg, s = cfg.num_bg_queries, cfg.mask_stride
    image_sizes = [tuple(x) for x in out["image_sizes"]]
    B = len(image_sizes)
    dev = out["pred_logits"].device
    out_sizes = [(int(x.get("height", sz[0])), int(x.get("width", sz[1]))) for x, sz in zip(batched_inputs, image_sizes)]
    if not do_postprocess:
        out_sizes_inst = image_sizes
    else:
        out_sizes_inst = out_sizes
    pmap = {1: [0]} if task == "grounding" else batched_inputs[0]["positive_map_label_to_token"]      # hipie_img.py:322-326
    C = len(pmap)
    is_thing = [x.get("is_thing", {}) for x in batched_inputs]
    has_thing = [any(t.values()) for t in is_thing]

    box_cls, box_pred = out["pred_logits"][:, nbg:].float(), out["pred_boxes"][:, nbg:].float().contiguous()
    iou = out["pred_boxious"][:, nbg:].float()
    pred_masks = out["pred_masks"][:, nbg:, 0]                                   # (B, Q, hm, wm)
    Q = box_cls.shape[1]
    logits = convert_grounding_to_od_logits(box_cls, C, pmap, is_thing, ["FG" if h else None for h in has_thing],
                                            cfg.mode_free, cfg.max_pool)        # (B, Q, C)
    if use_clip and cfg.ota:
        # hipie_img.py:592-609: the detector's class probabilities are fused with MaskCLIP's per-mask class probabilities
        if cfg.transform_eval and C > 1:
            p_det = F.softmax(logits.sigmoid() / cfg.pano_temp_fg, dim=-1)
        else:
            p_det = logits.sigmoid()
        # MaskCLIP resizes the image and the mask logits independently to its input size: the two must cover the same extent.  The
        # reference runs at B = 1, where the logits span the image's own padded canvas; in a mixed-size batch image i therefore gets the
        # logits of ITS canvas (size rounded up to the backbone's divisibility), not of the batch canvas -- the B = 1 result of that image.
        def own_canvas(i):
            hm = -(-image_sizes[i][0] // 32) * 32 // s
            wm = -(-image_sizes[i][1] // 32) * 32 // s
            return pred_masks[i][:, :min(hm, pred_masks.shape[-2]), :min(wm, pred_masks.shape[-1])]
        fused = torch.stack(_clip_logits_all(model, batched_inputs, out, [own_canvas(i) for i in range(B)], [p_det[i] for i in range(B)]))
        allowed = (logits[:, :1] != NEG).float()                                  # the reference's is_thing_mask (first query row)
        prob = torch.sqrt((fused.sigmoid() * allowed) ** cfg.clip_fg_a * iou.sigmoid() ** cfg.clip_fg_b)
    else:
        prob = torch.sqrt(logits.sigmoid() * iou.sigmoid())
    if cfg.ota:
        nms_scores, idxs = prob.max(2)
        keep, count = ops.batched_nms(box_pred, nms_scores.contiguous(), idxs.contiguous(), cfg.nms_thresh)
    else:
        if task == "detection" and not cfg.use_bg_for_pano:
            raise ValueError("OTA off needs TEST.USE_BG_FOR_PANO_ON (the reference has no NMS keep list in that branch)")
        keep = torch.arange(Q, device=dev, dtype=torch.int32).expand(B, Q).contiguous()
        count = torch.full((B,), Q, dtype=torch.int32, device=dev)
    keep_l = keep.long().clamp_min(0)
    rows_ok = torch.arange(Q, device=dev)[None, :] < count[:, None]             # (B, Q) kept rows, score order
    prob_k = torch.gather(prob, 1, keep_l[:, :, None].expand(-1, -1, C))
    prob_k = torch.where(rows_ok[:, :, None], prob_k, prob_k.new_tensor(-2.0))
    K = min(max_num_inst, Q * C)
    flat = prob_k.flatten(1)
    if flat.is_cuda and K <= 1024:
        top_i, top_v = ops.topk(flat.float().contiguous(), K, want_values=True)      # one launch; ties in index order
    else:
        top_v, top_i = torch.topk(flat, K, dim=1)                                    # host-side unit tests of this index logic / DETECTIONS_PER_IMAGE > 1024
    top_row, labels = torch.div(top_i, C, rounding_mode="floor"), top_i % C
    top_q = torch.gather(keep_l, 1, top_row)                                     # fg query index of every instance
    boxes = _cxcywh_to_xyxy(torch.gather(box_pred, 1, top_q[:, :, None].expand(-1, -1, 4)))
    wh = _cached(("wh", tuple(image_sizes), str(dev)),
                 lambda: torch.tensor([[w, h, w, h] for (h, w) in image_sizes], dtype=torch.float32, device=dev)[:, None, :])
    boxes = boxes * wh
    # segmentation_postprocess: scale to the output resolution, clip, drop empty boxes (ddetrs.py:1046-1063)
    sc = _cached(("sc", tuple(image_sizes), tuple(out_sizes_inst), str(dev)), lambda: torch.tensor(
        [[ow / w, oh / h, ow / w, oh / h] for (h, w), (oh, ow) in zip(image_sizes, out_sizes_inst)], dtype=torch.float32, device=dev)[:, None, :])
    lim = _cached(("lim", tuple(out_sizes_inst), str(dev)), lambda: torch.tensor(
        [[ow, oh, ow, oh] for (oh, ow) in out_sizes_inst], dtype=torch.float32, device=dev)[:, None, :])
    if do_postprocess:
        boxes = torch.minimum((boxes * sc).clamp_min(0), lim)
        nonempty = ((boxes[..., 2] - boxes[..., 0]) > 0) & ((boxes[..., 3] - boxes[..., 1]) > 0)
    else:
        nonempty = torch.ones(boxes.shape[:2], dtype=torch.bool, device=dev)
    n_inst = torch.minimum(count.long() * C, count.new_tensor(K).long())         # num_inst = min(max_num_inst, prob.numel())
    ok = (torch.arange(K, device=dev)[None, :] < n_inst[:, None]) & nonempty
    return {"boxes": boxes, "scores": top_v, "labels": labels, "query": top_q, "ok": ok, "keep": keep_l, "count": count, "logits": logits,
            "pred_masks": pred_masks, "image_sizes": image_sizes, "out_sizes": out_sizes, "out_sizes_inst": out_sizes_inst,
            "pmap": pmap, "C": C, "is_thing": is_thing, "task": task}


@torch.no_grad()
def inference_compact(model, out, batched_inputs, topk=100, do_postprocess=True):
    """the (B, topk, 7) block of parallel.compact_predictions(inference(..., with_masks=False, with_sem_pan=False), topk)
    -- [x0, y0, x1, y1, score, class, query index] per instance, score order, zero padded -- built entirely on the device:
    the instances are packed to the front of every row with a stable sort instead of a host round trip, so a data-parallel
    evaluation loop never synchronises with the host between the forward and the all-gather."""
    d = _instances_on_device(model, out, batched_inputs, do_postprocess)
    out.pop("_clip_states", None)
    ok = d["ok"]
    B, K = ok.shape
    order = torch.sort((~ok).to(torch.int8), dim=1, stable=True)[1]                  # instances first, original (score) order kept
    rows = torch.cat([d["boxes"], d["scores"][:, :, None], d["labels"][:, :, None].float(), d["query"][:, :, None].float()], 2)
    rows = torch.where(ok[:, :, None], rows, rows.new_zeros(()))
    rows = torch.gather(rows, 1, order[:, :, None].expand(-1, -1, rows.shape[2]))
    if K >= topk:
        return rows[:, :topk].contiguous()
    return torch.cat([rows, rows.new_zeros(B, topk - K, rows.shape[2])], 1)


@torch.no_grad()
def inference(model, out, batched_inputs, do_postprocess=True, with_masks=True, with_sem_pan=True):
    try:
        return _inference(model, out, batched_inputs, do_postprocess, with_masks, with_sem_pan)
    finally:
        out.pop("_clip_states", None)            # MaskCLIP's per-call image state (_clip_states) does not outlive the call


def _inference(model, out, batched_inputs, do_postprocess=True, with_masks=True, with_sem_pan=True):
    """a22 dictionary -> list of {"instances": Instances, "panoptic_seg": (label map, segments_info), "sem_seg"} like
    HIPIE_IMG.forward's eval branch (hipie_img.py:313-362).  with_masks / with_sem_pan False skip the instance masks /
    the semantic + panoptic maps (e.g. a detection-only consumer, or the compact all-gather block of parallel.py)."""
    d = _instances_on_device(model, out, batched_inputs, do_postprocess)
    cfg = model.cfg
    s = cfg.mask_stride
    boxes, top_v, labels, top_q, ok, keep_l, count, logits = (d[k] for k in ("boxes", "scores", "labels", "query", "ok", "keep", "count", "logits"))
    pred_masks, image_sizes, out_sizes, out_sizes_inst = d["pred_masks"], d["image_sizes"], d["out_sizes"], d["out_sizes_inst"]
    pmap, C, is_thing, task = d["pmap"], d["C"], d["is_thing"], d["task"]
    B = len(image_sizes)
    dev = boxes.device
    ok_h, count_h = ok.cpu(), None                                               # the one host sync of the instance path
    results = []
    tables = []
    for i in range(B):
        sel = torch.nonzero(ok_h[i]).flatten().to(dev)
        inst = Instances(out_sizes_inst[i])
        inst.pred_boxes = Boxes(boxes[i][sel])
        inst.scores = top_v[i][sel]
        inst.pred_classes = labels[i][sel]
        qidx = top_q[i][sel].to(torch.int32).contiguous()
        inst.query_index = qidx               # foreground-query index of every instance (the data-parallel gather carries it)
        if with_masks:
            inst.pred_masks = ops.mask_finalize(pred_masks[i].contiguous(), qidx, s, image_sizes[i], out_sizes_inst[i], cfg.mask_thres)
        res = {"instances": inst, "panoptic_seg": (None, None), "sem_seg": None}
        results.append(res)
    if task == "detection" and with_sem_pan:
        md_logits = out["pred_logits_maskdino"].float()
        md_masks = out["pred_masks_maskdino"]
        mode = None if (cfg.use_bg_for_pano or cfg.bg_cls_agnostic) else "BG"
        logits_bg = convert_grounding_to_od_logits(md_logits, C, pmap, is_thing, mode, cfg.mode_free, cfg.max_pool)
        if not cfg.use_bg_for_pano:
            count_h = count.tolist()
        cls_list, masks_list = [], []
        for i in range(B):
            if cfg.use_bg_for_pano:
                logits_all, masks_all = logits_bg[i], md_masks[i]
            else:
                k = keep_l[i, :count_h[i]]
                logits_all = torch.cat([logits[i][k], logits_bg[i]], 0)
                masks_all = torch.cat([pred_masks[i][k], md_masks[i].to(pred_masks.dtype)], 0)
            if cfg.transform_eval:
                cls_all = F.softmax(logits_all.sigmoid() / cfg.pano_temp, dim=-1)
            else:
                cls_all = logits_all.sigmoid()
            cls_list.append(cls_all)
            masks_list.append(masks_all)
        if getattr(model, "enable_clip", False):
            # hipie_img.py:731-747: the masks the reference hands to CLIP are the x4 up-sampled logits cropped to the image
            if _clip_states(model, batched_inputs, out)[0] == "batched":
                # the x4 up-sampling, the crop and MaskCLIP's resize to its input size as ONE pair of small operators per image geometry
                blocked = [model.clip.blocked_patches_upsampled(masks_list[i][None], s, image_sizes[i])[0] for i in range(B)]
                cls_list = [c.softmax(-1) for c in _clip_logits_all(model, batched_inputs, out, masks_list, cls_list, blocked=blocked)]
            else:
                ups = []
                for i in range(B):
                    up = F.interpolate(masks_list[i][:, None].float(), scale_factor=float(s), mode="bilinear", align_corners=False)
                    ups.append(up[:, 0, :image_sizes[i][0], :image_sizes[i][1]])
                cls_list = [c.softmax(-1) for c in _clip_logits_all(model, batched_inputs, out, ups, cls_list)]
                del ups
        for i in range(B):
            sem, tab = _sem_pan(cls_list[i], masks_list[i], s, image_sizes[i], out_sizes[i], thing_vector(is_thing[i], C, dev), cfg,
                                0 if getattr(getattr(model, "precision", None), "einsum", 0) in (0, 1, 4) else 1)
            results[i]["sem_seg"] = sem
            tables.append(tab)
        for i, info in enumerate(_segments_info(tables)):
            results[i]["panoptic_seg"] = (tables[i]["pan"], info)
    return results
