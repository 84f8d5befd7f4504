"""CPU: MaskCLIP score fusion -- the oracle (oracle/clip.py, oracle/post.py) and the product module's host path
(hipie_amd/open_vocab.py) against tests/golden/maskclip.npz, which was produced by the reference's OWN MaskCLIP /
HIPIE_IMG.get_clip_logits / HIPIE_IMG.inference (MODEL.CLIP.ENABLED on) over the open_clip stand-in of tests/golden/ref_shim.py."""
import pytest
import torch

import _synth
from oracle import clip as oc
from oracle import post as op
from util import Golden, rel_err

torch.set_grad_enabled(False)
TOL = 2e-5


def setup():
    g = Golden("maskclip")
    cfg = g.meta["clip_cfg"]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=95)
    test = [t["name"].split(",") for t in g.meta["test_labels"]]
    train = [t["name"].split(",") for t in g.meta["train_labels"]]
    return g, cfg, sd, test, train


def text_embed(cfg, sd, labels):
    flat = [t for syn in labels for t in syn]
    return oc.clip_encode_text(_synth.clip_tokenize(flat, cfg["context"], cfg["vocab"]), sd, "", cfg)


def test_oracle_mask_embed_text_embed_logits():
    g, cfg, sd, test, train = setup()
    labels = oc.prompt_labels_photo(test)
    te = text_embed(cfg, sd, labels)
    assert rel_err(te, g["text_embed"]) < TOL
    me = oc.get_mask_embed(g["image"], g["mask"], sd, "", cfg)
    assert rel_err(me, g["mask_embed"]) < TOL
    lg = oc.pred_logits(me, te, labels, sd, "")
    assert rel_err(lg, g["open_logits"]) < TOL
    assert lg.shape == (1, 7, len(test))


@pytest.mark.parametrize("mode", ["MUL", "ADD"])
def test_oracle_fusion(mode):
    g, cfg, sd, test, train = setup()
    ov = oc.category_overlap(test, train)
    assert ov.tolist() == [1, 1, 0, 1, 0]              # person / dog / sky overlap the training vocabulary; traffic light, zebra crossing do not
    fused = oc.get_clip_logits(g["open_logits"][0], g["pred_open_prob"], ov, 0.4, 0.45, mode)
    assert rel_err(fused, g["fused_" + mode]) < TOL


def clip_callable(g, cfg, sd, test, train, image01, agg="MUL"):
    labels = oc.prompt_labels_photo(test)
    te = text_embed(cfg, sd, labels)
    ov = oc.category_overlap(test, train)

    def f(i, mask_logits, prob):
        me = oc.get_mask_embed(image01[None], mask_logits[None], sd, "", cfg)
        return oc.get_clip_logits(oc.pred_logits(me, te, labels, sd, "")[0], prob, ov, 0.4, 0.45, agg)
    return f


def test_oracle_inference_with_clip_matches_reference():
    """HIPIE_IMG.inference with MODEL.CLIP.ENABLED (both call sites) -> instances, panoptic and semantic maps."""
    g, cfg, sd, test, train = setup()
    P = g.meta["post"]
    sizes = [tuple(s) for s in P["sizes"]]
    a22 = _synth.synth_a22(sizes, P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"])
    pmap = {int(k): v for k, v in g.meta["pmap"].items()}
    is_thing = {int(k): v for k, v in g.meta["is_thing"].items()}
    img = _synth.synth_images(sizes, seed=98)[0] / 255.0
    res = op.inference(a22, sizes, pmap, "detection", [is_thing], num_bg=P["n_bg"], clip=clip_callable(g, cfg, sd, test, train, img))
    inst = res[0]["instances"]
    assert torch.equal(inst["classes"].long(), g["post_classes"])
    assert rel_err(inst["scores"], g["post_scores"]) < 5e-5
    assert rel_err(inst["boxes"], g["post_boxes"]) < 1e-5
    pan, info = res[0]["panoptic_seg"]
    assert info == g.meta["segments"]
    assert torch.equal(pan.long(), g["post_panoptic"].long())
    assert rel_err(g.like("post_semseg", res[0]["sem_seg"].float()), g["post_semseg"]) < 5e-5


def test_product_maskclip_host_path_matches_reference():
    """hipie_amd.open_vocab.MaskCLIP on the CPU (same decomposition as the device path: mask tokens read the image tokens' keys /
    values through their patch masks; no (Q + T)^2 attention) against the reference's mask embeddings / logits / fused logits."""
    from hipie_amd.open_vocab import MaskCLIP, get_clip_logits
    g, cfg, sd, test, train = setup()
    m = MaskCLIP("tiny", cfg=cfg, tokenize=lambda t: _synth.clip_tokenize(t, cfg["context"], cfg["vocab"]))
    m.load_clip_state_dict(sd)
    assert len(m.state_dict()) == 0                                      # CLIP weights are not part of a HIPIE checkpoint
    labels = oc.prompt_labels_photo(test)
    te = m.build_text_embed(labels)
    assert rel_err(te, g["text_embed"]) < TOL
    out = m(g["image"], g["mask"], te, labels)
    assert rel_err(out["mask_embed"], g["mask_embed"]) < TOL
    assert rel_err(out["mask_pred_open_logits"], g["open_logits"]) < TOL
    for mode in ("MUL", "ADD"):
        fused = get_clip_logits(m, g["image"][0], g["mask"][0], test, train, g["pred_open_prob"], 0.4, 0.45, mode)
        assert rel_err(fused, g["fused_" + mode]) < TOL


def test_visibility_maps_without_the_upsampled_tensor():
    """MaskCLIP.blocked_patches_upsampled (the x4 up-sampling of the semantic / panoptic mask logits, the crop to the image and MaskCLIP's
    resize to its input size as one pair of small operators: hipie_img.py:731-747 + clip.py:299-321) gives the patch-visibility bits of the
    three-step route, on blob-shaped masks with ~45 % of the patches visible, for a crop that cuts the up-sampled map and one that does not."""
    import torch.nn.functional as F
    from hipie_amd.open_vocab import MaskCLIP
    cfg = dict(width=128, layers=2, heads=4, patch=14, image_size=112, embed_dim=64, text_width=64, text_layers=2, text_heads=4, context=16,
               vocab=100, quick_gelu=True)
    torch.manual_seed(0)
    m = MaskCLIP("tiny", cfg=cfg, tokenize=lambda t: None)
    mask = F.interpolate(torch.randn(2, 300, 6, 5) * 3 - 1.5, size=(50, 44), mode="bilinear", align_corners=False)
    for crop in ((200, 176), (187, 150)):
        up = F.interpolate(mask, scale_factor=4.0, mode="bilinear", align_corners=False)[:, :, :crop[0], :crop[1]]
        want = m.blocked_patches(up)
        got = m.blocked_patches_upsampled(mask, 4, crop)
        assert 0.3 < float((~want).float().mean()) < 0.7
        assert int((want != got).sum()) <= 2, int((want != got).sum())          # a logit within fp32 rounding of 0 may fall on the other side
