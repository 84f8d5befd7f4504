"""GPU: the post-processing row (hipie_amd/postprocess.py + hipie_batched_nms + hipie_mask_finalize) against the golden from
the reference's own HIPIE_IMG.inference and against the oracle (oracle/post.py) on larger random cases.
Integer results (NMS keep lists, classes, segment tables) are compared exactly; boolean masks / label maps may differ on a
<= 1e-4 fraction of pixels (a logit within 1 ulp of the threshold); float outputs within 1e-4 of the output scale."""
import types

import pytest
import torch

import _synth
from oracle import post as op
from test_post_oracle import check_against_golden, post_case
from util import Golden

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def fake_model(nbg, **kw):
    from hipie_amd.config import HipieConfig
    cfg = HipieConfig()
    cfg.num_bg_queries = nbg
    for k, v in kw.items():
        setattr(cfg, k, v)
    return types.SimpleNamespace(cfg=cfg)


def as_dict(res):
    out = []
    for r in res:
        i = r["instances"]
        out.append(dict(instances=dict(boxes=i.pred_boxes.tensor, scores=i.scores, classes=i.pred_classes, masks=i.pred_masks,
                                       image_size=i.image_size), panoptic_seg=r["panoptic_seg"], sem_seg=r["sem_seg"]))
    return out


def run_product(a22, sizes, pmap, is_thing, out_hw, kw, task, nbg):
    from hipie_amd.postprocess import inference
    dev = "cuda"
    out = {k: v.to(dev) for k, v in a22.items()}
    out["image_sizes"] = sizes
    batched = [{"task": task, "positive_map_label_to_token": pmap, "is_thing": is_thing, "height": out_hw[i][0],
                "width": out_hw[i][1]} for i in range(len(sizes))]
    return as_dict(inference(fake_model(nbg, **kw), out, batched))


@pytest.mark.parametrize("cname", ["default", "evalyaml", "grounding"])
def test_post_product_matches_reference_golden(cname):
    g = Golden("post")
    res = run_product(*post_case(g, cname))
    check_against_golden(g, cname, res, tol=1e-4, mask_mismatch=1e-4)


@pytest.mark.parametrize("B,Q,C,trick", [(3, 900, 80, None), (2, 1024, 7, None), (1, 1, 1, None), (2, 257, 1, 1), (2, 300, 5, 0)])
def test_batched_nms_bit_exact(B, Q, C, trick):
    from hipie_amd import ops
    g = torch.Generator().manual_seed(Q * 7 + C)
    centers = torch.rand(B, 12, 4, generator=g) * torch.tensor([0.8, 0.8, 0.5, 0.5]) + 0.1
    which = torch.randint(0, 12, (B, Q), generator=g)
    boxes = torch.gather(centers, 1, which[..., None].expand(-1, -1, 4)) + 0.03 * torch.randn(B, Q, 4, generator=g)
    boxes[..., 2:] = boxes[..., 2:].abs().clamp_min(1e-3)
    scores = torch.rand(B, Q, generator=g)
    scores[:, ::17] = scores[:, 0:1].clone()                                   # ties: the stable sort decides
    classes = torch.randint(0, C, (B, Q), generator=g)
    keep, count = ops.batched_nms(boxes.cuda(), scores.cuda(), classes.cuda(), 0.7, coordinate_trick=trick)
    for b in range(B):
        xyxy = op.box_cxcywh_to_xyxy(boxes[b])
        if trick is None:
            want = op.batched_nms(xyxy, scores[b], classes[b], 0.7)
            if Q * 4 > 4000:                                           # torchvision's per-class path returns score order too
                want = want[scores[b][want].sort(descending=True, stable=True)[1]]
        elif trick:
            mc = xyxy.max()
            want = op.nms(xyxy + (classes[b].float() * (mc + 1))[:, None], scores[b], 0.7)
        else:
            km = torch.zeros(Q, dtype=torch.bool)
            for c in classes[b].unique():
                ci = torch.where(classes[b] == c)[0]
                km[ci[op.nms(xyxy[ci], scores[b][ci], 0.7)]] = True
            order = scores[b].sort(descending=True, stable=True)[1]
            want = order[km[order]]
        n = int(count[b])
        assert n == want.numel()
        got = keep[b, :n].cpu().long()
        if trick is None and Q * 4 > 4000:
            assert sorted(got.tolist()) == sorted(want.tolist())       # equal scores may be ordered differently
        else:
            assert torch.equal(got, want)
        assert (keep[b, n:] == -1).all()


def test_batched_nms_rejects_oversize():
    from hipie_amd import ops
    from hipie_amd._lib import HipieLibraryError
    with pytest.raises((HipieLibraryError, RuntimeError)):
        ops.batched_nms(torch.rand(1, 1025, 4).cuda(), torch.rand(1, 1025).cuda(), torch.zeros(1, 1025, dtype=torch.long).cuda(), 0.7)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("crop,out", [((200, 256), (200, 256)), ((250, 131), (333, 97)), ((256, 256), (64, 511))])
def test_mask_finalize_matches_torch(dtype, crop, out):
    import torch.nn.functional as F
    from hipie_amd import ops
    g = torch.Generator().manual_seed(5)
    m = (torch.randn(13, 64, 64, generator=g) * 3).to(dtype).cuda()
    qidx = torch.tensor([12, 0, 5, 5, 7], dtype=torch.int32).cuda()
    got = ops.mask_finalize(m, qidx, 4, crop, out, 0.5)
    ref = F.interpolate(m[qidx.long()][:, None].float(), size=(256, 256), mode="bilinear", align_corners=False)
    ref = (ref.sigmoid() > 0.5)[:, :, :crop[0], :crop[1]]
    ref = F.interpolate(ref.float(), size=out, mode="nearest").squeeze(1).byte()
    assert got.shape == ref.shape and got.dtype == torch.uint8
    assert (got != ref).float().mean() < 2e-4
    full = ops.mask_finalize(m, None, 4, crop, out, 0.5)
    assert torch.equal(full[qidx.long()], got)


@pytest.mark.parametrize("use_bg", [True, False])
def test_post_product_matches_oracle_large(use_bg):
    """more queries / segments than the golden: stuff classes repeat, so the merge of stuff regions is exercised."""
    sizes = [(384, 512), (512, 448)]
    nbg, nfg, nmd, L, ncls = 10, 300, 120, 64, 9
    a22 = _synth.synth_a22(sizes, nbg, nfg, nmd, L, seed=123)
    a22["pred_logits_maskdino"] = a22["pred_logits_maskdino"] * 2.0
    _, _, pmap = _synth.synth_token_ids(2, ncls, L, seed=74)
    is_thing = {c + 1: (c % 3 == 0) for c in range(ncls)}
    kw = dict(use_bg_for_pano=use_bg, bg_cls_agnostic=not use_bg, max_pool=not use_bg, object_mask_threshold=0.2,
              overlap_threshold=0.5)
    out_hw = [(384, 512), (300, 333)]
    want = op.inference(a22, sizes, pmap, "detection", [is_thing] * 2, out_sizes=out_hw, num_bg=nbg, **kw)
    got = run_product(a22, sizes, pmap, is_thing, out_hw, kw, "detection", nbg)
    n_seg = 0
    for w, g_ in zip(want, got):
        wi, gi = w["instances"], g_["instances"]
        assert torch.equal(gi["classes"].cpu(), wi["classes"])
        assert torch.allclose(gi["scores"].cpu(), wi["scores"], atol=1e-5)
        assert torch.allclose(gi["boxes"].cpu(), wi["boxes"], atol=1e-3)
        assert (gi["masks"].cpu() != wi["masks"]).float().mean() < 1e-4
        assert g_["panoptic_seg"][1] == w["panoptic_seg"][1]
        assert (g_["panoptic_seg"][0].cpu() != w["panoptic_seg"][0]).float().mean() < 1e-4
        assert (g_["sem_seg"].cpu() - w["sem_seg"]).abs().max() < 1e-4 * w["sem_seg"].abs().max()
        n_seg += len(w["panoptic_seg"][1])
        stuff_ids = [s_["category_id"] for s_ in w["panoptic_seg"][1] if not s_["isthing"]]
        assert len(stuff_ids) == len(set(stuff_ids))
    assert n_seg >= 4


@pytest.mark.parametrize("N,C,crop,out,prec", [(300, 9, (200, 256), (200, 256), 0), (130, 80, (250, 131), (333, 97), 0),
                                               (77, 150, (256, 256), (256, 256), 1), (40, 33, (64, 100), (50, 300), 1),
                                               (60, 847, (128, 96), (128, 96), 0), (900, 1203, (96, 128), (120, 160), 1)])
def test_sem_pan_kernel_matches_torch_formulation(N, C, crop, out, prec):
    """hipie_sem_pan against the reference's tensor formulation (two bilinear resizes, sigmoid, einsum, argmax, areas)."""
    import torch.nn.functional as F
    from hipie_amd import ops
    g = torch.Generator().manual_seed(N + C)
    a22 = _synth.synth_a22([(256, 256)], 0, 1, N, 8, seed=N)
    masks = a22["pred_masks_maskdino"][0].cuda()                              # (N, 64, 64) blobs
    cls = torch.softmax(torch.randn(N, C, generator=g) * 3, -1).cuda()
    scores = cls.max(-1)[0]
    ps = torch.where(scores > 0.4, scores, scores.new_tensor(-1.0))
    sem, idx, own, area = ops.sem_pan(masks, cls, ps, 4, crop, out, prec)
    up = F.interpolate(masks[:, None], scale_factor=4.0, mode="bilinear", align_corners=False)[:, :, :crop[0], :crop[1]]
    if crop != out:
        up = F.interpolate(up, size=out, mode="bilinear", align_corners=False)
    sig = up[:, 0].sigmoid()
    want_sem = torch.einsum("qc,qhw->chw", cls, sig)
    tol = 3e-5 if prec == 0 else 6e-3
    assert (sem - want_sem).abs().max() < tol * want_sem.abs().max()
    kept = ps > 0
    w_ids = (torch.where(kept, ps, ps.new_tensor(-1.0)).view(-1, 1, 1) * sig).argmax(0)
    if kept.any():
        assert (idx.long() != w_ids).float().mean() < 1e-4
        w_own = sig.gather(0, w_ids[None])[0] >= 0.5
        assert (own != w_own).float().mean() < 1e-4
    else:
        assert (idx == -1).all() and not own.any()
    w_area = (sig >= 0.5).view(N, -1).sum(1)
    assert ((area.long() - w_area).abs() <= 2 + w_area // 2000).all()       # a logit within an ulp of 0 may flip a pixel


def test_sem_pan_no_query_kept():
    from hipie_amd import ops
    masks = torch.randn(20, 16, 16).cuda()
    cls = torch.full((20, 5), 0.2).cuda()
    sem, idx, own, area = ops.sem_pan(masks, cls, torch.full((20,), -1.0).cuda(), 4, (64, 64), (64, 64), 0)
    assert (idx == -1).all() and not own.any() and torch.isfinite(sem).all()


@pytest.mark.parametrize("task,topk", [("detection", 100), ("detection", 7), ("grounding", 100)])
def test_inference_compact_equals_host_path(task, topk):
    """the device-only compact block of the data-parallel step == compact_predictions(inference(...)): same instances, same
    order, same zero padding -- including images whose clipped boxes go empty (dropped in the middle of the score order)."""
    from hipie_amd import parallel
    from hipie_amd.postprocess import inference, inference_compact
    sizes = [(384, 512), (512, 448), (256, 256)]
    nbg, nfg, nmd, L, ncls = 10, 300, 120, 64, 9
    a22 = _synth.synth_a22(sizes, nbg, nfg, nmd, L, seed=321)
    a22["pred_boxes"][1, nbg:nbg + 150, 2:] = 0.0                      # zero-size boxes: non-empty filter drops them after scaling
    _, _, pmap = _synth.synth_token_ids(3, ncls, L, seed=74)
    is_thing = {c + 1: (c % 3 == 0) for c in range(ncls)}
    out = {k: v.cuda() for k, v in a22.items()}
    out["image_sizes"] = sizes
    batched = [{"task": task, "positive_map_label_to_token": pmap, "is_thing": is_thing, "height": 300 + 10 * i, "width": 333}
               for i in range(len(sizes))]
    model = fake_model(nbg)
    want = parallel.compact_predictions(inference(model, out, batched, with_masks=False, with_sem_pan=False), topk=topk)
    got = inference_compact(model, out, batched, topk=topk)
    assert got.shape == want.shape == (3, topk, parallel.PRED_FIELDS)
    assert torch.equal(got, want)
    if task == "detection" and topk == 100:
        assert int((want[1, :, 4] > 0).sum()) < int((want[0, :, 4] > 0).sum())        # the empty boxes of image 1 were dropped


def _clip_setup(policy_split):
    from hipie_amd.modeling.transformer import set_split
    from hipie_amd.open_vocab import MaskCLIP
    g = Golden("maskclip")
    cfg = g.meta["clip_cfg"]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=95)
    m = MaskCLIP("tiny", cfg=cfg, tokenize=lambda t: _synth.clip_tokenize(t, cfg["context"], cfg["vocab"]))
    m.load_clip_state_dict(sd)
    m = m.cuda().eval()
    set_split(m, policy_split)
    return g, m


@pytest.mark.parametrize("split", [True, False])
def test_maskclip_device_path(split):
    """MaskCLIP on the device (image tokens on hipie_flash_attn, mask tokens through their patch masks, linears on hipie_gemm's split
    operands or the fp32 library) against the reference's mask embeddings / per-mask logits / fused logits (tests/golden/maskclip.npz)."""
    from hipie_amd.open_vocab import get_clip_logits, prompt_labels_photo
    from util import rel_err
    g, m = _clip_setup(split)
    test = [t["name"].split(",") for t in g.meta["test_labels"]]
    train = [t["name"].split(",") for t in g.meta["train_labels"]]
    labels = prompt_labels_photo(test)
    te = m.build_text_embed(labels)
    assert rel_err(te.cpu(), g["text_embed"]) < 1e-4
    out = m(g["image"].cuda(), g["mask"].cuda(), te, labels)
    e1, e2 = rel_err(out["mask_embed"].cpu(), g["mask_embed"]), rel_err(out["mask_pred_open_logits"].cpu(), g["open_logits"])
    fused = get_clip_logits(m, g["image"][0].cuda(), g["mask"][0].cuda(), test, train, g["pred_open_prob"].cuda(), 0.4, 0.45, "MUL")
    e3 = rel_err(fused.cpu(), g["fused_MUL"])
    print("maskclip device path (split=%s): mask_embed %.1e open_logits %.1e fused %.1e" % (split, e1, e2, e3))
    assert e1 < 1e-3 and e2 < 1e-3 and e3 < 1e-3          # the image-token attention runs on fp16 operands


def test_maskclip_full_size():
    """f-2 at the size the eval yamls run it (MODEL.CLIP.NAME ViT-L-14-336: width 1024, 24 layers, 577 image tokens) with the mask-token
    count of one image of the shipped model (910 detection + 300 MaskDINO queries = 1210): random CLIP weights, one 1024 x 1024 image,
    blob-shaped mask logits.  The device path (mask tokens as extra rows reading the image tokens' keys through their patch masks; split
    GEMM linears, image-token attention on hipie_flash_attn) against the oracle's restatement of MaskCLIP.get_mask_embed with the full
    (Q + 577)^2 boolean attention mask (oracle/clip.py <- hipie/open_vocab/clip.py:258-353) on the CPU: 1e-3 on the mask embeddings."""
    import time
    from oracle import clip as oc
    from hipie_amd.modeling.transformer import set_split
    from hipie_amd.open_vocab import CLIP_CONFIGS, MaskCLIP
    from util import rel_err
    torch.manual_seed(11)
    m = MaskCLIP("ViT-L-14-336", tokenize=lambda t: None)
    cfg = CLIP_CONFIGS["ViT-L-14-336"]
    sd = {k: v.detach().clone() for k, v in torch.nn.Module.state_dict(m.clip).items()}
    m.loaded = True
    m = m.cuda().eval()
    set_split(m, True)
    gen = torch.Generator().manual_seed(12)
    Q = 1210
    image = torch.rand(1, 3, 1024, 1024, generator=gen)
    coarse = torch.randn(1, Q, 12, 12, generator=gen) * 3.0 - 1.5           # blobs: every mask token sees its own subset of the 24 x 24 patches
    mask = torch.nn.functional.interpolate(coarse, size=(256, 256), mode="bilinear", align_corners=False)
    got = m.get_mask_embed(image.cuda(), mask.cuda())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = m.get_mask_embed(image.cuda(), mask.cuda())
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    want = oc.get_mask_embed(image, mask, sd, "", cfg)
    t_cpu = time.perf_counter() - t0
    err = rel_err(got.float().cpu(), want)
    blocked = (torch.nn.functional.max_pool2d(torch.nn.functional.interpolate(mask, size=(336, 336), mode="bilinear", align_corners=False).sigmoid(), 14, 14) < 0.5)
    print("maskclip FULL SIZE (ViT-L/14-336, Q = %d): device %.1f ms, oracle on the CPU %.1f s, rel err %.1e; patches blocked per token %.0f%%"
          % (Q, t_dev * 1e3, t_cpu, err, 100 * float(blocked.float().mean())))
    assert got.shape == (1, Q, cfg["embed_dim"]) and err < 1e-3
    assert 0.2 < float(blocked.float().mean()) < 0.95                        # the masks really restrict the attention


def test_post_product_with_maskclip_matches_reference():
    """postprocess.inference with MODEL.CLIP.ENABLED (both call sites of the fusion, hipie_img.py:592-609 and :735-747) against
    the reference's own HIPIE_IMG.inference run with its MaskCLIP: classes exactly, scores / boxes / semantic map to 1e-3,
    the panoptic map up to pixels whose winning score is within rounding of a competitor's."""
    from hipie_amd.postprocess import inference
    from util import rel_err
    g, m = _clip_setup(True)
    P = g.meta["post"]
    sizes = [tuple(s) for s in P["sizes"]]
    a22 = _synth.synth_a22(sizes, P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"])
    pmap = {int(k): v for k, v in g.meta["pmap"].items()}
    is_thing = {int(k): v for k, v in g.meta["is_thing"].items()}
    img = _synth.synth_images(sizes, seed=98)[0]
    model = fake_model(P["n_bg"], clip_alpha=0.4, clip_beta=0.45, clip_agg_mode="MUL", clip_fg_a=0.3, clip_fg_b=1.7, pano_temp_fg=0.06)
    model.enable_clip, model.clip, model.train_labels = True, m, g.meta["train_labels"]
    out = {k: v.cuda() for k, v in a22.items()}
    out["image_sizes"] = sizes
    batched = [{"task": "detection", "positive_map_label_to_token": pmap, "is_thing": is_thing, "image": img.cuda(),
                "open_seg_labels": g.meta["test_labels"]}]
    r = inference(model, out, batched)[0]
    inst = r["instances"]
    # the 100 instances include (query, class) pairs of classes the FG mode disallows: probability exactly 0 -- torch.topk orders
    # those ties differently on the host and on the device.  Everything with a positive score is compared in order, the zero tail as a set.
    pos = g["post_scores"] > 0
    n = int(pos.sum())
    assert bool(pos[:n].all()) and n > 50
    assert torch.equal(inst.pred_classes.cpu().long()[:n], g["post_classes"][:n])
    assert rel_err(inst.scores.cpu(), g["post_scores"]) < 1e-3
    assert rel_err(inst.pred_boxes.tensor.cpu()[:n], g["post_boxes"][:n]) < 1e-4
    assert sorted(inst.pred_classes.cpu().tolist()[n:]) == sorted(g["post_classes"].tolist()[n:])
    pan, info = r["panoptic_seg"]
    assert info == g.meta["segments"]
    assert (pan.cpu().long() != g["post_panoptic"].long()).float().mean() < 1e-3
    assert rel_err(g.like("post_semseg", r["sem_seg"].cpu().float()), g["post_semseg"]) < 1e-3


def test_post_maskclip_mixed_sizes_match_single_image_runs():
    """MODEL.CLIP.ENABLED in a MIXED-size batch: MaskCLIP resizes image and mask logits independently, so image i must be scored with
    the logits of its own canvas -- every image of the batch gets the result of its own B = 1 run (the only case the reference has)."""
    from hipie_amd.postprocess import inference
    from util import rel_err
    g, m = _clip_setup(True)
    P = g.meta["post"]
    sizes = [(96, 160), (160, 96)]
    a22 = _synth.synth_a22(sizes, P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"] + 1)
    pmap = {int(k): v for k, v in g.meta["pmap"].items()}
    is_thing = {int(k): v for k, v in g.meta["is_thing"].items()}
    imgs = _synth.synth_images(sizes, seed=97)
    model = fake_model(P["n_bg"], clip_alpha=0.4, clip_beta=0.45, clip_agg_mode="MUL", clip_fg_a=0.3, clip_fg_b=1.7, pano_temp_fg=0.06)
    model.enable_clip, model.clip, model.train_labels = True, m, g.meta["train_labels"]
    s = model.cfg.mask_stride

    def item(i):
        return {"task": "detection", "positive_map_label_to_token": pmap, "is_thing": is_thing, "image": imgs[i].cuda(),
                "open_seg_labels": g.meta["test_labels"]}
    out = {k: v.cuda() for k, v in a22.items()}
    out["image_sizes"] = sizes
    both = inference(model, out, [item(0), item(1)])
    assert out["pred_masks"].shape[-2:] == (160 // s, 160 // s)
    for i in range(2):
        hm, wm = -(-sizes[i][0] // 32) * 32 // s, -(-sizes[i][1] // 32) * 32 // s
        one = {k: v[i:i + 1].cuda() for k, v in a22.items()}
        one["pred_masks"] = one["pred_masks"][..., :hm, :wm].contiguous()
        one["pred_masks_maskdino"] = one["pred_masks_maskdino"][..., :hm, :wm].contiguous()
        one["image_sizes"] = [sizes[i]]
        alone = inference(model, one, [item(i)])[0]["instances"]
        got = both[i]["instances"]
        n = int((alone.scores > 0).sum())
        assert n > 20
        assert torch.equal(got.pred_classes[:n], alone.pred_classes[:n])
        assert rel_err(got.scores.cpu(), alone.scores.cpu()) < 1e-4
        assert rel_err(got.pred_boxes.tensor[:n].cpu(), alone.pred_boxes.tensor[:n].cpu()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,Nq,Nk,hd", [(2, 16, 150, 577, 64), (1, 3, 5, 33, 32), (3, 2, 130, 64, 64)])
def test_attn_f32_rows_mask_per_query_row(B, H, Nq, Nk, hd):
    """hipie_attn_f32_rows / hipie_attn_split_rows (the mask tokens' attention of MaskCLIP: every query row has its own set of visible keys) against
    softmax(masked_fill(q k^T, -inf)) v in double; q / k / v as column blocks of one projection output (strided views); key 0 always visible."""
    from hipie_amd import ops
    g = torch.Generator().manual_seed(B * 7 + Nq)
    qkv = torch.randn(B, max(Nq, Nk), 3, H, hd, generator=g, dtype=torch.float64)
    q, k, v = qkv[:, :Nq, 0], qkv[:, :Nk, 1], qkv[:, :Nk, 2]
    vis = torch.rand(B, Nq, Nk, generator=g) < 0.3
    vis[:, :, 0] = True
    sc = (q.transpose(1, 2) * hd ** -0.5) @ k.permute(0, 2, 3, 1)
    want = (sc.masked_fill(~vis[:, None], float("-inf")).softmax(-1) @ v.transpose(1, 2)).transpose(1, 2).reshape(B, Nq, H * hd)
    dq = qkv.float().cuda()
    for split, tol in ((False, 2e-6), (True, 4e-6)):
        got = ops.attn_f32_rows(dq[:, :Nq, 0], dq[:, :Nk, 1], dq[:, :Nk, 2], hd ** -0.5, vis.cuda(), split=split)
        err = float((got.cpu().double() - want).abs().max() / want.abs().max())
        assert err < tol, (split, err)
