"""CPU: the oracle (oracle/) against the golden fixtures produced by the reference's own modules.
This is what pins the oracle (there are no reference tests for this path beyond ops/test.py)."""
import pytest
import torch

import _synth
from oracle import model as om
from oracle import ops as oo
from util import Golden, rel_err

torch.set_grad_enabled(False)
TOL = 2e-5   # fp32 CPU vs fp32 CPU, different summation order only


def test_msda_reference_recipe():
    g = Golden("msda")
    for tag, dt, tol in (("ref_double", torch.float64, 1e-12), ("ref_float", torch.float32, 1e-6)):
        out = oo.ms_deform_attn_core(g[tag + "_value"].to(dt), g[tag + "_shapes"], g[tag + "_loc"].to(dt), g[tag + "_attn"].to(dt))
        assert rel_err(out, g[tag + "_out"]) < tol


def msda_case(g, tag):
    m = g.meta[tag]
    shapes = g[tag + "_shapes"]
    S = int(shapes.prod(1).sum())
    gen = torch.Generator().manual_seed(m["seed"])
    value = torch.randn(m["B"], S, 8, 32, generator=gen)
    loc = torch.rand(m["B"], m["Lq"], 8, 4, 4, 2, generator=gen) * 1.2 - 0.1
    attn = torch.softmax(torch.randn(m["B"], m["Lq"], 8, 16, generator=gen), -1).view(m["B"], m["Lq"], 8, 4, 4)
    return value, shapes, loc, attn


@pytest.mark.parametrize("tag", ["hot_enc", "hot_dec", "hot_rect"])
def test_msda_hot_geometry(tag):
    g = Golden("msda")
    value, shapes, loc, attn = msda_case(g, tag)
    out = oo.ms_deform_attn_core(value, shapes, loc, attn)
    assert rel_err(g.like(tag + "_out", out), g[tag + "_out"]) < TOL


def vit_attn_case(g, name):
    c = g.meta["cases"][name]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in c["manifest"].items()}, seed=21)
    x = _synth.synth_tensor("x_" + name, c["x_shape"], seed=22) * 8.0
    return c, sd, x


@pytest.mark.parametrize("name", ["window14", "global16", "global64", "global_rect", "global84", "global128"])
def test_vit_attention(name):
    g = Golden("vit_attn")
    c, sd, x = vit_attn_case(g, name)
    out = oo.vit_attention(x, sd, "", c["heads"])
    assert rel_err(g.like(name + "_out", out), g[name + "_out"]) < TOL


def test_vit_backbone():
    g = Golden("vit_backbone")
    cfg = g.meta["cfg"]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=31)
    x = _synth.synth_tensor("vit_in", g.meta["x_shape"], seed=32)
    out = om.vit_backbone(x, sd, "", cfg)
    for k in ("res3", "res4", "res5"):
        assert rel_err(g.like(k, out[k]), g[k]) < TOL


def bi_case(g, name):
    c = g.meta["cases"][name]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in c["manifest"].items()}, seed=41)
    v = _synth.synth_tensor("v_" + name, (c["B"], c["Nv"], 256), seed=42) * 16 * c["scale"]
    l = _synth.synth_tensor("l_" + name, (c["B"], c["L"], 768), seed=43) * 27 * c["scale"]
    return c, sd, v, l, g[name + "_mask"]


@pytest.mark.parametrize("name", ["L20", "L600_pad", "clamp"])
def test_bi_attention(name):
    g = Golden("bi_attn")
    c, sd, v, l, mask = bi_case(g, name)
    ov, ol = oo.bi_attention_block(v, l, mask, sd, "")
    assert rel_err(g.like(name + "_v", ov), g[name + "_v"]) < TOL
    assert rel_err(g.like(name + "_l", ol), g[name + "_l"]) < TOL


def test_bert_short_and_chunked():
    g = Golden("bert")
    cfg = g.meta["cfg"]
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=51)
    for tag in ("short", "long"):
        out = om.bert_encoder(g[tag + "_ids"], g[tag + "_mask"], sd, "model.", cfg)["hidden"]
        assert rel_err(g.like(tag + "_hidden", out), g[tag + "_hidden"]) < 5e-5


def dyn_case(g, name):
    c = g.meta["cases"][name]
    feats = _synth.synth_tensor("dm_feats_" + name, (c["B"], 8, c["H"], c["W"]), seed=61) * 20
    params = _synth.synth_tensor("dm_params_" + name, (1, c["B"] * c["Q"], 169), seed=63) * 6
    return c, feats, g[name + "_refs"], params


@pytest.mark.parametrize("name", ["sq", "rect"])
def test_dynamic_mask(name):
    g = Golden("dynamic_mask")
    c, feats, refs, params = dyn_case(g, name)
    out = oo.dynamic_mask(feats, refs, params, [c["Q"]] * c["B"], stride=8, up=2)
    assert rel_err(g.like(name + "_out", out), g[name + "_out"]) < TOL


def test_aligned_bilinear():
    g = Golden("dynamic_mask")
    x = _synth.synth_tensor("ab_x", (3, 1, 9, 13), seed=64) * 10
    assert rel_err(oo.aligned_bilinear(x, 2), g["aligned_bilinear_out"]) < 1e-6


def e2e_inputs(g, task):
    cfg = g.meta["cfg"]
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist"))
    imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
    ids, mask, pmap = _synth.synth_token_ids(2, g.meta[task]["n_classes"], g.meta[task].get("max_len", 64), seed=74,
                                             pad_to=g.meta[task].get("pad_to"))
    return cfg, sd, imgs, ids, mask


E2E_KEYS = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino",
            "pred_logits_maskdino", "pred_boxes_maskdino"]


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_tiny(task):
    """the full a22 dictionary of DDETRSegmUniDN.coco_inference, free-running top-k (same fp32 CPU arithmetic,
    so the selections agree) and with pinned indices."""
    g = Golden("e2e_tiny")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, task)
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    assert rel_err(g.like(task + "_lang_hidden", lang["hidden"]), g[task + "_lang_hidden"]) < 5e-5
    out = om.coco_inference(imgs, lang, sd, cfg, task=task, topk_fg=g[task + "_topk_fg"], topk_md=g[task + "_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like(task + "_" + k, out[k]), g[task + "_" + k]) < 2e-4, k
    free = om.coco_inference(imgs, lang, sd, cfg, task=task)
    assert torch.equal(free["topk_fg"], g[task + "_topk_fg"])
    assert torch.equal(free["topk_md"], g[task + "_topk_md"])


def test_e2e_full_size():
    """BASELINE.json's headline configuration itself (full ViT-H, shipped head sizes, one 1024 x 1024 image): the oracle against the
    reference's own coco_inference -- pins the checker, and bench.py's `cpu_baseline`, at the size the metric is quoted on (~1.5 min
    of CPU; big outputs compared on the fixture's strided subsample)."""
    g = Golden("e2e_full")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, "detection")
    n = len(g.meta["sizes"])
    lang = om.bert_encoder(ids[:n], mask[:n], sd, "text_encoder.body.model.", cfg)
    assert rel_err(g.like("detection_lang_hidden", lang["hidden"]), g["detection_lang_hidden"]) < 5e-5
    out = om.coco_inference(imgs, lang, sd, cfg, task="detection", topk_fg=g["detection_topk_fg"], topk_md=g["detection_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like("detection_" + k, out[k]), g["detection_" + k]) < 2e-4, k


def test_e2e_long_prompt():
    """BASELINE configs[3]-style prompt through the full path: 815 tokens (BertEncoder's > 512 chunker) padded to 896."""
    g = Golden("e2e_long_tiny")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, "detection")
    assert ids.shape[1] == 896 and int(mask.sum(1).max()) > 512
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    assert rel_err(g.like("detection_lang_hidden", lang["hidden"]), g["detection_lang_hidden"]) < 5e-5
    out = om.coco_inference(imgs, lang, sd, cfg, task="detection", topk_fg=g["detection_topk_fg"], topk_md=g["detection_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like("detection_" + k, out[k]), g["detection_" + k]) < 2e-4, k


def test_e2e_pad_max_4096():
    """MODEL.LANGUAGE_BACKBONE.PAD_MAX at MAX_QUERY_LEN 4096 (the shipped eval yamls; hipie_img.py:904-909): a 9-class caption padded to 4096
    tokens through the > 512 branch of BertEncoder -- the hidden states of the padding stay ZERO (bert_model.py:118-127), fusion and class
    logits run over 4096 mostly-masked columns.  Also pins what the product's trimming relies on: every padding row of the reference's
    language stream and every padding column of its class logits is the same row / column."""
    g = Golden("e2e_padmax_tiny")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, "detection")
    assert ids.shape[1] == 4096 and int(mask.sum(1).max()) < 64
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    n_real = int(mask.sum(1).max())
    assert float(lang["hidden"][:, n_real:].abs().max()) == 0.0
    assert rel_err(g.like("detection_lang_hidden", lang["hidden"]), g["detection_lang_hidden"]) < 5e-5
    out = om.coco_inference(imgs, lang, sd, cfg, task="detection", topk_fg=g["detection_topk_fg"], topk_md=g["detection_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like("detection_" + k, out[k]), g["detection_" + k]) < 2e-4, k
    for k in ("pred_logits", "pred_logits_maskdino"):
        pad = out[k][..., n_real:]
        assert float((pad - pad[..., :1]).abs().max()) <= 1e-5 * float(out[k].abs().max()), k


def test_e2e_full_c80_bench_inputs():
    """the workload bench.py times (BASELINE configs[2]: ViT-H, 1024 x 1024, image 0 of bench.synth_batch, 80-class caption of 194 tokens,
    and the separate grounding call): the oracle against the reference's own coco_inference on those inputs (~2 min of CPU per task: the
    grounding half only with HIPIE_SLOW_TESTS=1, to keep the CPU suite short)."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "e2e_full_c80.npz")):
        pytest.skip("tests/golden/e2e_full_c80.npz not generated")
    import bench
    g = Golden("e2e_full_c80")
    bi = g.meta["bench_inputs"]
    cfg = g.meta["cfg"]
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist"))
    for task in ("detection", "grounding") if os.environ.get("HIPIE_SLOW_TESTS") == "1" else ("detection",):
        b = bench.synth_batch(None, 1, bi["size"], bi["n_classes"], bi["L"], "cpu", seed=bi["seed"], task=task)[0]
        ids, mask = b["input_ids"][None], b["attention_mask"][None]
        assert ids.shape[1] == g.meta[task]["L"]
        lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
        assert rel_err(g.like(task + "_lang_hidden", lang["hidden"]), g[task + "_lang_hidden"]) < 5e-5
        out = om.coco_inference([b["image"]], lang, sd, cfg, task=task, topk_fg=g[task + "_topk_fg"], topk_md=g[task + "_topk_md"])
        for k in E2E_KEYS:
            assert rel_err(g.like(task + "_" + k, out[k]), g[task + "_" + k]) < 2e-4, (task, k)


def test_e2e_r50_512_literal_config0():
    """BASELINE configs[0] LITERALLY: R50 with the shipped head sizes, ONE 512 x 512 image, ONE referring expression -- the oracle against the
    reference's own CPU run (tests/golden/gen_golden.py::gen_e2e_r50_512)."""
    g = Golden("e2e_r50_512")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, "grounding")
    assert len(imgs) == 1 and tuple(imgs[0].shape[-2:]) == (512, 512) and g.meta["grounding"]["n_classes"] == 1
    ids, mask = ids[:1], mask[:1]                                # one prompt row per image
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    out = om.coco_inference(imgs, lang, sd, cfg, task="grounding", topk_fg=g["grounding_topk_fg"], topk_md=g["grounding_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like("grounding_" + k, out[k]), g["grounding_" + k]) < 2e-4, k


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_r50_tiny(task):
    """BASELINE configs[0] (R50, one text prompt: grounding) / [1] (R50, class prompts): a22 against the reference run behind its own
    detectron2 ResNet-50."""
    g = Golden("e2e_r50_tiny")
    cfg, sd, imgs, ids, mask = e2e_inputs(g, task)
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    out = om.coco_inference(imgs, lang, sd, cfg, task=task, topk_fg=g[task + "_topk_fg"], topk_md=g[task + "_topk_md"])
    for k in E2E_KEYS:
        assert rel_err(g.like(task + "_" + k, out[k]), g[task + "_" + k]) < 2e-4, k


def test_stages_tiny():
    g = Golden("stages_tiny")
    e = Golden("e2e_tiny")
    cfg, sd, imgs, ids, mask = e2e_inputs(e, "detection")
    lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    out = om.coco_inference(imgs, lang, sd, cfg, task="detection", want_stages=True)
    st = out["_stages"]
    for i, n in enumerate(["res3", "res4", "res5"]):
        assert rel_err(g.like("feat%d" % i, st["feats"][n]), g["feat%d" % i]) < TOL
        assert rel_err(g.like("pos%d" % i, st["poses"][i]), g["pos%d" % i]) < TOL
        assert torch.equal(g.like("mask%d" % i, st["fmasks"][i]), g["mask%d" % i])
    assert rel_err(g.like("memory", st["memory"]), g["memory"]) < 1e-4
    assert rel_err(g.like("vl0_lang", st["lang_hidden"]), g["vl0_lang"]) < 1e-4
    assert rel_err(g.like("dec_hs", st["hs"]), g["dec_hs"]) < 1e-4
    assert rel_err(g.like("dec_refs", st["inter"]), g["dec_refs"]) < 1e-4
    assert rel_err(g.like("mask_head_out", st["mask_head"]), g["mask_head_out"]) < 1e-4
    assert rel_err(g.like("md_enc_memory", st["md_enc_memory"]), g["md_enc_memory"]) < 1e-4
    assert rel_err(g.like("md_mask_features", st["md_mask_features"]), g["md_mask_features"]) < 1e-4
    for i in range(4):
        assert rel_err(g.like("md_ms%d" % i, st["md_ms"][i]), g["md_ms%d" % i]) < 1e-4


def test_resnet50_backbone():
    g = Golden("resnet50")
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=81)
    x = _synth.synth_tensor("r50_in", g.meta["x_shape"], seed=82) * 2
    out = om.resnet50_backbone(x, sd, "")
    for k in ("res3", "res4", "res5"):
        assert rel_err(g.like(k, out[k]), g[k]) < 5e-5


@pytest.mark.parametrize("case", ["recipe30", "recipe32", "recipe64", "recipe71", "recipe1025", "hot"])
def test_msda_backward_oracle(case):
    """autograd through the oracle's explicit corner-gather formulation == the gradients of the reference's own differentiable core
    (F.grid_sample + autograd, ops/functions/ms_deform_attn_func.py:41-62) in double: the checker of hipie_msda_backward."""
    g = Golden("msda_bwd")
    tag, D = [(t, d) for n, t, d in g.meta["cases"] if n == case][0]
    value, shapes, loc, attn, gout = _synth.msda_bwd_inputs(tag, D)
    with torch.enable_grad():
        value.requires_grad_(True), loc.requires_grad_(True), attn.requires_grad_(True)
        out = oo.ms_deform_attn_core(value, shapes, loc, attn)
        gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), gout)
    assert rel_err(out.detach(), g[case + "_out"]) < 1e-12
    assert rel_err(gv, g[case + "_gvalue"]) < 1e-12
    assert rel_err(gl, g[case + "_gloc"]) < 1e-10
    assert rel_err(ga, g[case + "_gattn"]) < 1e-12



def test_refinit_distribution_table_covers_the_model_and_is_reproducible():
    """tests/golden/refinit_stats.json (statistics of the reference's OWN initialisation, measured by gen_golden.py on the reference's
    constructors) has an entry for every floating-point tensor of the full ViT-H model; drawing from it is deterministic, zero biases /
    unit LayerNorm scales stay constants, the tensors the reference zero-initialises that would make a kernel trivial are N(0, 0.02)."""
    g = Golden("e2e_full_refinit")
    assert g.meta.get("dist") == "refinit"
    table = _synth.refinit_stats()
    man = {k: tuple(v) for k, v in g.meta["manifest"].items()}
    missing = []
    for k, shp in man.items():
        pre = "detr." if k.startswith("detr.") else "text_encoder.body."
        ck = pre + (_synth.canonical_key(k[len(pre):]) if pre == "detr." else k[len(pre):])
        if len(shp) and "position_ids" not in k and "token_type_ids" not in k and "num_batches" not in k and ck not in table:
            missing.append(k)
    assert not missing, missing[:5]
    keys = ["detr.detr.backbone.0.backbone.blocks.0.attn.qkv.weight", "detr.detr.backbone.0.backbone.blocks.0.attn.qkv.bias",
            "detr.detr.backbone.0.backbone.blocks.0.attn.rel_pos_h", "detr.detr.transformer.encoder.layers.0.linear1.weight",
            "detr.detr.transformer.encoder.layers.0.self_attn.sampling_offsets.bias"]
    sub = {k: man[k] for k in keys}
    a = _synth.synth_full_state_dict(sub, dist="refinit")
    b = _synth.synth_full_state_dict(sub, dist="refinit")
    for k in keys:
        assert torch.equal(a[k], b[k])
    assert abs(float(a[keys[0]].std()) - 0.02) < 1e-3                 # trunc_normal_(std=0.02) (backbone/vit.py:350)
    assert float(a[keys[1]].abs().max()) == 0.0                       # zero bias
    assert abs(float(a[keys[2]].std()) - 0.02) < 4e-3                 # zero in the reference -> N(0, 0.02)
    assert abs(float(a[keys[3]].std()) - (2.0 / (256 + 2048)) ** 0.5) < 2e-3      # xavier_uniform_ (deformable_transformer_dino.py:109-112)
    assert torch.allclose(a[keys[4]], _synth.msda_grid_bias().view_as(a[keys[4]]))
