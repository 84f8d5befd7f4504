"""CPU: host-side logic of the product (no kernel launches): state_dict compatibility with the reference, config,
preprocessing, chunked text encoder."""
import os

import pytest
import torch

import _synth
from util import Golden, rel_err

torch.set_grad_enabled(False)


def tiny_model(device="cpu"):
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    g = Golden("e2e_tiny")
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    return g, cfg, HIPIE_IMG(cfg, Precision.parity(), device=device)


def test_state_dict_matches_reference_manifest():
    """every key/shape of the reference model's state_dict (DetectionCheckpointer must load unchanged)."""
    g, cfg, model = tiny_model()
    ours = {k: list(v.shape) for k, v in model.state_dict().items()}
    ref = g.meta["manifest"]
    assert sorted(ours) == sorted(ref)
    for k in ref:
        assert ours[k] == ref[k], k


def test_tied_parameters_are_shared_like_the_reference():
    g, cfg, model = tiny_model()
    d = model.detr
    assert d.detr.transformer.decoder.bbox_embed is d.detr.bbox_embed
    assert d.detr.transformer.decoder.class_embed[0] is d.detr.class_embed[0]
    p = d.mask_dino.predictor
    assert p.bbox_embed[0] is p._bbox_embed and p.decoder.bbox_embed[1] is p._bbox_embed and p.decoder.norm is p.decoder_norm


def test_full_size_vit_huge_key_count():
    from hipie_amd.config import HipieConfig
    c = HipieConfig.vit_huge()
    assert (c.vit_embed_dim, c.vit_depth, c.vit_heads) == (1280, 32, 16) and c.vit_window_blocks == [0, 1, 3, 4, 6, 7, 9, 10]
    assert c.backbone_channels == [640, 1280, 1280]


def test_preprocess_matches_oracle():
    from oracle import model as om
    g, cfg, model = tiny_model()
    imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
    il = model.preprocess_image([{"image": im} for im in imgs])
    from hipie_amd.modeling.transformer import nested_tensor_from_images
    nt = nested_tensor_from_images(list(il), 32)
    t, m, sizes = om.preprocess(imgs, g.meta["cfg"])
    assert torch.equal(nt.mask, m) and rel_err(nt.tensors, t) < 1e-6 and il.image_sizes == sizes


def test_text_encoder_matches_golden_cpu():
    """the BERT wrapper is pure torch (no custom kernel): check short and chunked paths against the reference golden."""
    from hipie_amd.config import HipieConfig
    from hipie_amd.modeling.text import BertEncoder
    g = Golden("bert")
    enc = BertEncoder(HipieConfig.from_dict(g.meta["cfg"])).eval()
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=51)
    enc.load_state_dict(sd, strict=True)
    for tag in ("short", "long"):
        out = enc({"input_ids": g[tag + "_ids"], "attention_mask": g[tag + "_mask"]}, sep=1012)["hidden"]
        assert rel_err(g.like(tag + "_hidden", out), g[tag + "_hidden"]) < 5e-5


def test_library_missing_fails_loudly(monkeypatch):
    from hipie_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libhipie_mi355.so")
    with pytest.raises(_lib.HipieLibraryError):
        _lib.load()


def test_msda_shim_installs_reference_module_name():
    import sys
    from hipie_amd import msda_shim
    mod = msda_shim.install()
    import MultiScaleDeformableAttention as MSDA
    assert MSDA is mod and callable(MSDA.ms_deform_attn_forward) and callable(MSDA.ms_deform_attn_backward)
    # both entry points refuse host tensors loudly (there is no CPU path behind the plugin)
    v = torch.zeros(1, 4, 1, 2)
    shapes, start = torch.tensor([[2, 2]]), torch.tensor([0])
    loc, w = torch.zeros(1, 1, 1, 1, 1, 2), torch.ones(1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_forward(v, shapes, start, loc, w, 64)
    with pytest.raises(RuntimeError):
        MSDA.ms_deform_attn_backward(v, shapes, start, loc, w, torch.zeros(1, 1, 2), 64)
    sys.modules.pop("MultiScaleDeformableAttention")


def test_resnet50_state_dict_and_cpu_forward_match_reference():
    """R50 is plain torch convolutions (no custom kernel), so its product module can be checked on the CPU against the
    golden produced by the reference's own detectron2 resnet.py."""
    from hipie_amd.config import Precision
    from hipie_amd.modeling.resnet import ResNet50
    g = Golden("resnet50")
    m = ResNet50(Precision.parity()).eval()
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert ours == g.meta["manifest"]
    m.load_state_dict(_synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=81))
    x = _synth.synth_tensor("r50_in", g.meta["x_shape"], seed=82) * 2
    out = m(x)
    for k in ("res3", "res4", "res5"):
        assert rel_err(g.like(k, out[k]), g[k]) < 5e-5
    out = m.cast_weights()(x)                  # FrozenBN folded into the convolutions (what finalize() runs with)
    for k in ("res3", "res4", "res5"):
        assert rel_err(g.like(k, out[k]), g[k]) < 5e-5


def test_r50_model_builds_with_reference_key_prefixes():
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    cfg = HipieConfig.r50()
    cfg.enc_layers = cfg.dec_layers = cfg.md_enc_layers = cfg.md_dec_layers = 1
    cfg.bert_layers = 1
    m = HIPIE_IMG(cfg, Precision.parity(), device="cpu")
    keys = m.state_dict().keys()
    assert "detr.detr.backbone.0.backbone.stem.conv1.norm.running_var" in keys
    assert "detr.detr.backbone.0.backbone.res5.2.conv3.weight" in keys
    assert m.state_dict()["detr.detr.input_proj.2.0.weight"].shape == (256, 2048, 1, 1)
    assert m.state_dict()["detr.mask_dino.pixel_decoder.adapter_1.weight"].shape == (256, 512, 1, 1)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_panoptic_merge_without_loops_matches_reference_loop(seed):
    """the loop-free panoptic merge (cumulative-sum ids + first-occurrence stuff merging) of hipie_amd/postprocess.py is
    plain torch, so it is checked here on the CPU against the oracle's restatement of panoptic_inference's loop."""
    import types
    from oracle import post as op
    from hipie_amd import postprocess as pp
    N, C = 60, 4
    a22 = _synth.synth_a22([(128, 160)], 0, 1, N, 8, seed=200 + seed)
    masks = a22["pred_masks_maskdino"][0]                                  # (N, 32, 40)
    g = torch.Generator().manual_seed(seed)
    cls_all = torch.softmax(torch.randn(N, C, generator=g) * 3, -1)
    is_thing = {1: True, 2: False, 3: False, 4: (seed % 2 == 0)}
    cfg = types.SimpleNamespace(object_mask_threshold=0.3, overlap_threshold=0.6)
    sem, tab = pp._sem_pan(cls_all, masks, 4, (128, 160), (100, 150), pp.thing_vector(is_thing, C, "cpu"), cfg)
    info = pp._segments_info([tab])[0]
    up = torch.nn.functional.interpolate(masks[:, None], scale_factor=4.0, mode="bilinear", align_corners=False)
    up = torch.nn.functional.interpolate(up[:, :, :128, :160], size=(100, 150), mode="bilinear", align_corners=False)[:, 0]
    pan_w, info_w = op.panoptic_inference(cls_all, up, is_thing, 0.3, 0.6)
    assert info == info_w
    assert torch.equal(tab["pan"], pan_w)
    assert torch.allclose(sem, op.semantic_inference(cls_all, up), atol=1e-5)
    stuff = [s for s in info_w if not s["isthing"]]
    assert len(info_w) >= 3 and len(stuff) >= 1
    MERGES.append(int(tab["valid"].sum()) - len(info_w))
    if seed == 3:
        assert sum(MERGES) >= 1            # some stuff segment re-used an earlier id of its class


MERGES = []


def test_config_from_yacs_key_mapping():
    """HipieConfig.from_yacs reads the reference's yacs keys (projects/HIPIE/hipie/config.py:150-257 + the eval yaml); a
    namespace with those key names stands in for the CfgNode (yacs / detectron2 are not installed here)."""
    from types import SimpleNamespace as NS
    from hipie_amd.config import HipieConfig
    cfg = _eval_cfg(**{"MODEL.BACKBONE.NAME": "D2ViT", "MODEL.VIT.NAME": "ViT-huge", "MODEL.DDETRS.TWO_STAGE_NUM_PROPOSALS": 900,
                       "MODEL.DDETRS.TWO_STAGE_NUM_BG_PROPOSALS": 10, "MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN": 4096,
                       "TEST.USE_BG_FOR_PANO_ON": False, "TEST.BG_CLS_AGNOSTIC": True, "TEST.MAX_POOL": True})
    m = cfg.MODEL
    c = HipieConfig.from_yacs(cfg)
    assert (c.backbone, c.vit_embed_dim, c.vit_depth, c.vit_heads) == ("vit", 1280, 32, 16)
    assert (c.num_queries, c.num_bg_queries, c.mask_stride, c.ctrl_layers) == (900, 10, 4, 3)
    assert (c.use_bg_for_pano, c.bg_cls_agnostic, c.max_pool) == (False, True, True)
    assert (c.pano_temp, c.overlap_threshold, c.object_mask_threshold, c.mask_thres) == (0.06, 0.8, 0.25, 0.5)
    assert (c.max_query_len, c.pad_max, c.clip_enabled) == (4096, True, False)
    m.BACKBONE.NAME = "build_resnet_backbone"
    md = NS(MODEL=NS(MaskDINO=NS(NUM_OBJECT_QUERIES=300, DEC_LAYERS=9, DIM_FEEDFORWARD=2048),
                     SEM_SEG_HEAD=NS(TRANSFORMER_ENC_LAYERS=6, DIM_FEEDFORWARD=2048, MASK_DIM=256, CONVS_DIM=256)))
    c = HipieConfig.from_yacs(cfg, md)
    assert c.backbone == "r50" and c.backbone_channels == [512, 1024, 2048]
    assert (c.md_num_queries, c.md_dec_layers, c.md_enc_layers) == (300, 9, 6)


def test_prompt_construction_matches_reference(tmp_path):
    """caption + positive_map_label_to_token (hipie_amd/prompts.py) against the outputs of the reference mapper's own
    functions on the same categories and tokenizer (tests/golden/gen_golden.py prompts)."""
    from hipie_amd import prompts
    g = Golden("prompts")
    assert g.meta["categories"] == _synth.PROMPT_CATEGORIES and g.meta["vocab"] == _synth.PROMPT_VOCAB
    tok = _synth.prompt_tokenizer(tmp_path)
    for things_only in (False, True):
        caption, pmap = prompts.create_queries_and_maps(_synth.PROMPT_CATEGORIES, tok, things_only=things_only)
        assert caption == g.meta["caption_%d" % things_only]
        assert {str(k): v for k, v in pmap.items()} == g.meta["pmap_%d" % things_only]
    item = prompts.detection_inputs(torch.zeros(3, 32, 32), _synth.PROMPT_CATEGORIES, tok)
    assert item["task"] == "detection" and item["input_ids"].shape == item["attention_mask"].shape
    assert item["is_thing"][8] is False and item["positive_map_label_to_token"][2] == [3, 4]


@pytest.mark.parametrize("max_pool", [False, True])
@pytest.mark.parametrize("mode", [None, "FG", "BG"])
def test_token_to_class_pooling_matches_reference_loop(mode, max_pool):
    """convert_grounding_to_od_logits as one GEMM / one padded gather (hipie_amd/postprocess.py) against the oracle's
    restatement of the reference's per-class loop (hipie_img.py:1025-1052), per-image thing maps included."""
    from oracle import post as op
    from hipie_amd import postprocess as pp
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(3, 17, 64, generator=g)
    _, _, pmap = _synth.synth_token_ids(2, 9, 64, seed=74)
    things = [{c: (c % 2 == 0) for c in pmap}, {c: True for c in pmap}, {}]
    got = pp.convert_grounding_to_od_logits(logits, len(pmap), pmap, things, mode, False, max_pool)
    for b in range(3):
        want = op.convert_grounding_to_od_logits(logits[b:b + 1], len(pmap), pmap, is_thing=things[b], mode=mode,
                                                 max_pool=max_pool)[0]
        assert torch.allclose(got[b], want, atol=1e-6)


def test_geometry_cache_keys_and_eviction():
    from hipie_amd.modeling import transformer as T
    calls = []

    def build(tag):
        calls.append(tag)
        return torch.tensor([len(calls)])
    a = T.geo_cached(("k", 1), "x", lambda: build("a"))
    b = T.geo_cached(("k", 1), "x", lambda: build("b"))           # hit
    c = T.geo_cached(("k", 2), "x", lambda: build("c"))           # other geometry
    d = T.geo_cached(None, "x", lambda: build("d"))               # no key: never cached
    assert calls == ["a", "c", "d"] and a is b and c is not a and d is not a
    own = T.collections.OrderedDict()
    for i in range(40):
        T.geo_cached(("own", i), "y", lambda: build("o"), store=own)
    assert len(own) <= 33 and ("own", 39) in [k[0] for k in own]
    imgs = [torch.zeros(3, 40, 50), torch.zeros(3, 33, 64)]
    nt1 = T.nested_tensor_from_images(imgs)
    nt2 = T.nested_tensor_from_images([torch.ones(3, 40, 50), torch.ones(3, 33, 64)])
    assert nt1.mask is nt2.mask and nt1.geo_key == nt2.geo_key and tuple(nt1.tensors.shape) == (2, 3, 64, 64)
    assert bool(nt1.mask[0, 39, 49]) is False and bool(nt1.mask[0, 40, 0]) is True and bool(nt1.mask[1, 0, 63]) is False
    nt3 = T.nested_tensor_from_images([torch.zeros(3, 40, 51), torch.zeros(3, 33, 64)])
    assert nt3.geo_key != nt1.geo_key and nt3.mask is not nt1.mask


def test_full_size_state_dict_matches_the_reference_checkpoint_layout():
    """the shipped configuration (ViT-H, 6+6 layers, 900+10 queries, MaskDINO 300 queries / 9 layers, BERT-base): every one of
    the 1624 state_dict entries of the reference's model (tests/golden/manifest_vit_huge.json, generated from the reference's
    own classes) exists here with the same shape, and nothing else -- a released checkpoint loads with strict=True."""
    import json
    import os
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest_vit_huge.json")))
    with torch.device("meta"):                       # shapes only: no 3 GB allocation
        m = HIPIE_IMG(HipieConfig.vit_huge(), Precision.parity(), device="cpu")
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert len(ref) == 1624
    assert sorted(ours) == sorted(ref)
    assert all(ours[k] == ref[k] for k in ref)
    assert sum(int(torch.tensor(v).prod()) for v in ours.values()) == 809749203
    # and the ResNet-50 configuration of BASELINE configs[0]/[1] (1436 entries)
    ref = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "manifest_r50.json")))
    with torch.device("meta"):
        m = HIPIE_IMG(HipieConfig.r50(), Precision.parity(), device="cpu")
    ours = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert len(ref) == 1436 and ours == ref


# ------------------------------------------------------------------------------------------------ registry boundary (b1)
def _fake_detectron2(monkeypatch):
    """the slice of detectron2 that hipie_amd.d2_registry touches, as in-memory modules (no detectron2 on any box here)."""
    import sys
    import types

    class Registry(object):
        def __init__(self, name):
            self._name, self._obj_map = name, {}

        def register(self, obj):
            assert obj.__name__ not in self._obj_map
            self._obj_map[obj.__name__] = obj
            return obj

        def get(self, name):
            return self._obj_map[name]

        def __contains__(self, name):
            return name in self._obj_map

    mods = {n: types.ModuleType(n) for n in ("detectron2", "detectron2.modeling", "detectron2.utils", "detectron2.utils.registry")}
    mods["detectron2.utils.registry"].Registry = Registry
    for r in ("BACKBONE_REGISTRY", "META_ARCH_REGISTRY", "SEM_SEG_HEADS_REGISTRY"):
        setattr(mods["detectron2.modeling"], r, Registry(r))
    for n, m in mods.items():
        monkeypatch.setitem(sys.modules, n, m)
    monkeypatch.delitem(sys.modules, "MultiScaleDeformableAttention", raising=False)
    return mods["detectron2.modeling"]


def _yacs_like(d):
    import types
    return types.SimpleNamespace(**{k: _yacs_like(v) if isinstance(v, dict) else v for k, v in d.items()})


def _eval_cfg(**over):
    """the keys HipieConfig.from_yacs reads, at the values of configs/eval/image_joint_r50_pan_maskdino_ade_test.yaml
    (hipie/config.py for the rest), shrunk where size does not matter for the test."""
    model = dict(
        DEVICE="cpu", PARALLEL_DET=False, DECOUPLE_TGT=True, STILL_TGT_FOR_BOTH=True, USE_IOU_BRANCH=True, STILL_CLS_FOR_ENCODER=True,
        LANG_GUIDE_DET=True, OTA=True, MODE_FREE_MATCHING_INFERENCE=False, PANO_TRANSFORM_EVAL=True, PANO_TEMPERATURE=0.06,
        OVERLAP_THRESHOLD=0.8, OBJECT_MASK_THRESHOLD=0.25, PIXEL_MEAN=[123.675, 116.28, 103.53], PIXEL_STD=[58.395, 57.12, 57.375],
        BACKBONE=dict(NAME="build_resnet_backbone"), VIT=dict(NAME="ViT-Base"),
        DDETRS=dict(HIDDEN_DIM=256, NHEADS=8, DIM_FEEDFORWARD=64, ENC_LAYERS=1, DEC_LAYERS=1, NUM_FEATURE_LEVELS=4, ENC_N_POINTS=4,
                    DEC_N_POINTS=4, TWO_STAGE_NUM_PROPOSALS=20, TWO_STAGE_NUM_BG_PROPOSALS=2, NUM_VL_LAYERS=1, VL_HIDDEN_DIM=2048,
                    MASK_STRIDE=4, CTRL_LAYERS=3, MASK_THRES=0.5, USE_DINO=True, TWO_STAGE=True, MIXED_SELECTION=True,
                    LOOK_FORWARD_TWICE=True, BG_QUERY_FROM_LANG=False, NEW_MASK_HEAD=False, USE_RAFT=False, USE_REL_COORD=True),
        LANGUAGE_BACKBONE=dict(LANG_DIM=768, MAX_QUERY_LEN=8192, PAD_MAX=True), DYHEAD=dict(LOG_SCALE=0.0, PRIOR_PROB=0.01),
        CLIP=dict(ENABLED=False, ENABLED_TRAIN=False, NAME="ViT-B-32", ALPHA=0.35, BETA=0.7, FG_IOU_A=0.3, FG_IOU_B=1.7, AGG_MODE="MUL"),
        PANO_TEMPERATURE_CLIP_FG=0.06, MASKDINO=dict(CONFIG_PATH="unused", ENABLED=True, SHARE_CLS_HEAD=False, FIXED_LINEAR_HEAD=False),
        MASK_ON=True, RESNETS=dict(DEPTH=50, STRIDE_IN_1X1=False))
    cfg = dict(MODEL=model, TEST=dict(USE_BG_FOR_PANO_ON=True, BG_CLS_AGNOSTIC=False, MAX_POOL=False))
    for path, v in over.items():
        node = cfg
        keys = path.split(".")
        for k in keys[:-1]:
            node = node[k]
        node[keys[-1]] = v
    return _yacs_like(cfg)


def _md_cfg(cfg):
    return _yacs_like(dict(MODEL=dict(MaskDINO=dict(NUM_OBJECT_QUERIES=12, DEC_LAYERS=1, DIM_FEEDFORWARD=64),
                                      SEM_SEG_HEAD=dict(TRANSFORMER_ENC_LAYERS=1, DIM_FEEDFORWARD=64, MASK_DIM=256, CONVS_DIM=256))))


def test_d2_registry_registers_and_builds_from_a_yacs_cfg(monkeypatch):
    """register() against a (fake) detectron2 puts the five classes under the reference's names; HIPIE_IMG(cfg)
    builds from a yacs-like CfgNode without touching weights (finalize is lazy: train_net.py loads the checkpoint AFTER
    build_model), honours MAX_QUERY_LEN / PAD_MAX, and refuses switch positions the build does not implement."""
    import sys
    import pytest
    from hipie_amd import d2_registry
    from hipie_amd.config import Precision
    modeling = _fake_detectron2(monkeypatch)
    out = d2_registry.register(Precision.parity(), md_cfg_loader=_md_cfg)
    assert "HIPIE_IMG" in modeling.META_ARCH_REGISTRY and "D2ViT" in modeling.BACKBONE_REGISTRY
    assert "MaskDINOHead" in modeling.SEM_SEG_HEADS_REGISTRY and "MaskDINOEncoder" in modeling.SEM_SEG_HEADS_REGISTRY
    assert "MaskDINODecoder" in out["TRANSFORMER_DECODER_REGISTRY"]
    assert "MultiScaleDeformableAttention" in sys.modules                       # the op shim of SURVEY 8b2
    d2_registry.register(Precision.parity(), md_cfg_loader=_md_cfg)              # idempotent

    model = modeling.META_ARCH_REGISTRY.get("HIPIE_IMG")(_eval_cfg())
    assert model.cfg.backbone == "r50" and model.cfg.max_query_len == 8192 and model.cfg.pad_max is True
    assert model.cfg.num_queries == 20 and model.cfg.md_num_queries == 12
    assert model._final is False                                                 # nothing weight-derived built at construction
    conv = model.detr.detr.backbone[0].backbone.stem.conv1
    assert conv.folded is None
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["detr.detr.backbone.0.backbone.stem.conv1.weight"].fill_(0.25)
    model._final = True
    model.load_state_dict(sd, strict=True)
    assert model._final is False                                                 # a checkpoint load re-arms finalize()
    # MODEL.CLIP.ENABLED (on in 10 of the 11 shipped eval yamls, with ALPHA 0.4 / BETA 0.45): the model constructs with its
    # MaskCLIP towers, which are not part of the state_dict, and a strict HIPIE checkpoint load does not ask for them
    mc = modeling.META_ARCH_REGISTRY.get("HIPIE_IMG")(_eval_cfg(**{"MODEL.CLIP.ENABLED": True, "MODEL.CLIP.ALPHA": 0.4, "MODEL.CLIP.BETA": 0.45}))
    assert mc.enable_clip and (mc.cfg.clip_alpha, mc.cfg.clip_beta, mc.cfg.clip_agg_mode, mc.cfg.clip_name) == (0.4, 0.45, "MUL", "ViT-B-32")
    assert not any(k.startswith("clip.") for k in mc.state_dict())
    assert sum(p.numel() for p in mc.clip.parameters()) > 1e8                    # ViT-B/32 towers are really there
    mc.load_state_dict(sd, strict=True)                                          # the CLIP-less checkpoint of the model above
    with pytest.raises(NotImplementedError, match="PARALLEL_DET"):
        modeling.META_ARCH_REGISTRY.get("HIPIE_IMG")(_eval_cfg(**{"MODEL.PARALLEL_DET": True}))
    with pytest.raises(NotImplementedError, match="USE_DINO"):
        modeling.META_ARCH_REGISTRY.get("HIPIE_IMG")(_eval_cfg(**{"MODEL.DDETRS.USE_DINO": False}))
    vit = modeling.BACKBONE_REGISTRY.get("D2ViT")(_eval_cfg(**{"MODEL.BACKBONE.NAME": "D2ViT"}), None)
    assert vit.blocks[0].attn.qkv.weight.shape == (3 * 768, 768)
    head = modeling.SEM_SEG_HEADS_REGISTRY.get("MaskDINOHead")(_eval_cfg())
    assert any(k.startswith("pixel_decoder.") for k in head.state_dict()) and any(k.startswith("predictor.") for k in head.state_dict())


def test_r50_fold_follows_a_checkpoint_load():
    """FrozenBN folds must come from the LOADED weights -- build, load_state_dict, finalize on the CPU and
    compare the folded stem convolution with conv + FrozenBN of the loaded parameters."""
    import torch
    from hipie_amd.modeling.resnet import ResNet50
    from hipie_amd.config import Precision
    torch.manual_seed(0)
    m = ResNet50(Precision.parity()).eval()
    m.cast_weights()                                                             # a fold of the random init (what finalize-at-build did)
    stale = m.stem.conv1.folded[0].clone()
    sd = {k: torch.randn_like(v) * 0.1 + (1.0 if k.endswith("running_var") else 0.0) for k, v in m.state_dict().items()}
    sd = {k: v.abs() + 0.5 if k.endswith("running_var") else v for k, v in sd.items()}
    m.load_state_dict(sd)
    m.cast_weights()
    assert not torch.allclose(stale, m.stem.conv1.folded[0])
    x = torch.randn(1, 3, 32, 32)
    c = m.stem.conv1
    want = c.norm(torch.nn.functional.conv2d(x, c.weight, None, stride=2, padding=3))
    got = c(x)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-5)


def test_predictor_input_recipe(tmp_path):
    """HIPIEPredictor.build_inputs (projects/HIPIE/predictor.py:324-371): output shape of ResizeShortestEdge on the sizes its
    docstring-era users know (COCO 480x640 -> 800x1067; a 1:3.3 panorama is bound by MAX_SIZE_TEST), BGR -> RGB, CHW float32,
    the dict keys of both tasks.  (fvcore is absent here, so detectron2's transform classes cannot be imported to pin this:
    the arithmetic below is the reference's formula, the resampling is the same PIL call.)"""
    import numpy as np
    from hipie_amd.predictor import HIPIEPredictor, resize_shortest_edge_shape
    assert resize_shortest_edge_shape(480, 640, 800, 1333) == (800, 1067)
    assert resize_shortest_edge_shape(640, 480, 800, 1333) == (1067, 800)
    assert resize_shortest_edge_shape(427, 640, 800, 1333) == (800, 1199)
    assert resize_shortest_edge_shape(300, 1000, 800, 1333) == (400, 1333)
    assert resize_shortest_edge_shape(1024, 1024, 1024, 1024) == (1024, 1024)

    class Echo(object):
        tokenizer = None

        def __call__(self, batched):
            return [batched[0]]
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (30, 45, 3), dtype=np.uint8)
    p = HIPIEPredictor(Echo(), min_size_test=60, max_size_test=80)
    out = p(img, "grounding", expressions="the left cat")
    assert out["image"].shape == (3, 53, 80) and out["image"].dtype == torch.float32       # 60/30 = 2 -> 90 > 80: bound by max
    assert (out["height"], out["width"], out["task"], out["expressions"], out["is_thing"]) == (30, 45, "grounding", "the left cat", {1: True})
    same = HIPIEPredictor(Echo(), min_size_test=30, max_size_test=45)(img, "grounding", expressions="x")
    assert torch.equal(same["image"], torch.as_tensor(img[:, :, ::-1].copy()).permute(2, 0, 1).float())      # identity resize, BGR -> RGB
    tok = _synth.prompt_tokenizer(tmp_path)                     # the small WordPiece vocabulary of the prompts golden
    cats = _synth.PROMPT_CATEGORIES[:3]
    det = HIPIEPredictor(Echo(), tokenizer=tok, min_size_test=30, max_size_test=45)(img, "detection", test_categories=cats)
    from hipie_amd import prompts
    caption, pmap = prompts.create_queries_and_maps(cats, tok)
    assert det["expressions"] == caption and det["positive_map_label_to_token"] == pmap
    assert det["is_thing"] == {i + 1: bool(c.get("isthing", 1)) for i, c in enumerate(cats)}
    with pytest.raises(ValueError):
        p(img, "sot")


@pytest.mark.parametrize("task,topk", [("detection", 100), ("detection", 7), ("grounding", 100)])
def test_inference_compact_equals_host_path_cpu(task, topk):
    """postprocess.inference_compact (device-side packing with a stable sort) == compact_predictions(inference(...)) -- on the CPU
    with OTA off (the NMS kernel is the only GPU-only piece of that path), including images whose clipped boxes go empty."""
    import types
    from hipie_amd import parallel
    from hipie_amd.config import HipieConfig
    from hipie_amd.postprocess import inference, inference_compact
    sizes = [(384, 512), (512, 448), (256, 256)]
    nbg, nfg, nmd, L, ncls = 10, 120, 40, 64, 9
    a22 = _synth.synth_a22(sizes, nbg, nfg, nmd, L, seed=321)
    a22["pred_boxes"][1, nbg:nbg + 60, 2:] = 0.0
    _, _, pmap = _synth.synth_token_ids(3, ncls, L, seed=74)
    is_thing = {c + 1: (c % 3 == 0) for c in range(ncls)}
    cfg = HipieConfig()
    cfg.num_bg_queries, cfg.ota, cfg.use_bg_for_pano = nbg, False, True
    model = types.SimpleNamespace(cfg=cfg)
    out = dict(a22)
    out["image_sizes"] = sizes
    batched = [{"task": task, "positive_map_label_to_token": pmap, "is_thing": is_thing, "height": 300 + 10 * i, "width": 333}
               for i in range(len(sizes))]
    want = parallel.compact_predictions(inference(model, out, batched, with_masks=False, with_sem_pan=False), topk=topk)
    got = inference_compact(model, out, batched, topk=topk)
    assert got.shape == want.shape == (3, topk, parallel.PRED_FIELDS)
    assert torch.equal(got, want)


# ------------------------------------------------------------------------------------------------ WordPiece tokenizer
def _wordpiece_vocab(tmp_path):
    import os
    import string
    words = ["person", "bicycle", "car", "motor", "##cycle", "air", "##plane", "traffic", "light", "fire", "hydrant", "stop", "sign", "hot",
             "dog", "potted", "plant", "tv", "wine", "glass", "hair", "dr", "##ier", "teddy", "bear", "the", "a", "of", "photo", "cafe",
             "naive", "tooth", "##brush", "##s", "##ing", "##ed", "un", "##able", "sky", "other", "merged", "zebra", "cross", "##ing"]
    toks = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(string.punctuation) + list(string.ascii_lowercase) + list(string.digits)
    toks += ["##" + c for c in string.ascii_lowercase + string.digits] + ["中", "文"]
    seen, out = set(), []
    for t in toks + words:
        if t not in seen:
            seen.add(t)
            out.append(t)
    path = os.path.join(str(tmp_path), "vocab.txt")
    with open(path, "w", encoding="utf-8") as f:
        f.write("\n".join(out) + "\n")
    return path


WORDPIECE_CASES = [
    "person. bicycle. car. motorcycle. airplane. traffic light. fire hydrant. stop sign. hot dog. potted plant. tv. wine glass. hair drier. teddy bear. toothbrush",
    "The Café of a naïve ZEBRA-crossing!!  (merged)", "a photo of a dog.", "", "   ", "unknown€symbol and 中文 text", "x" * 120 + " car",
    "tabs\tand\nnewlines\r\n. end", "İstanbul Ǆ ß ﬁ", "semi;colon:and,comma.dot?", "a.b.c", "sky-other-merged", "don't stop",
]


def _check_tokenizer_equal(own, hf, text, **kw):
    a = own(text, **kw)
    b = hf(text, return_offsets_mapping=True, return_tensors="pt", **kw)
    assert a.input_ids.tolist() == b["input_ids"].tolist(), text
    assert a.attention_mask.tolist() == b["attention_mask"].tolist(), text
    n = int(b["attention_mask"][0].sum())
    assert [tuple(o) for o in a.offsets[0][:n]] == [tuple(o) for o in b["offset_mapping"][0][:n].tolist()], text
    for c in range(len(text) + 1):
        try:
            want = b.char_to_token(c)
        except Exception:
            want = "err"
        if want != "err":
            assert a.char_to_token(c) == want, (text, c)


def test_wordpiece_tokenizer_matches_huggingface(tmp_path):
    """hipie_amd.tokenizer.BertWordPiece == transformers.BertTokenizerFast on the same vocabulary: ids, masks, offsets,
    char_to_token for every character, truncation and both paddings (the three call patterns of the reference)."""
    from transformers import BertTokenizerFast
    from hipie_amd.tokenizer import BertWordPiece
    vf = _wordpiece_vocab(tmp_path)
    own, hf = BertWordPiece(vf), BertTokenizerFast(vocab=vf, do_lower_case=True)
    for text in WORDPIECE_CASES:
        _check_tokenizer_equal(own, hf, text)
    _check_tokenizer_equal(own, hf, WORDPIECE_CASES[0], max_length=16, truncation=True)
    _check_tokenizer_equal(own, hf, WORDPIECE_CASES[2], max_length=24, padding="max_length", truncation=True)
    # a batch (forward_text, hipie_img.py:904-909): padding="longest" and "max_length", special-token mask
    batch = [WORDPIECE_CASES[0], WORDPIECE_CASES[2], WORDPIECE_CASES[1]]
    for pad, ml in (("longest", 64), ("max_length", 40)):
        a = own.batch_encode_plus(batch, max_length=ml, padding=pad, truncation=True, return_special_tokens_mask=True)
        b = hf(batch, max_length=ml, padding=pad, truncation=True, return_special_tokens_mask=True, return_tensors="pt")   # transformers 5 dropped batch_encode_plus
        assert a.input_ids.tolist() == b["input_ids"].tolist() and a.attention_mask.tolist() == b["attention_mask"].tolist()
        assert a.special_tokens_mask.tolist() == b["special_tokens_mask"].tolist()
    assert own(".").input_ids[0, 1] == hf(".").input_ids[1]                       # the separator id BertEncoder's chunker uses
    # the class-prompt map through the own tokenizer == through HuggingFace's
    from hipie_amd.prompts import create_queries_and_maps
    cats = _synth.PROMPT_CATEGORIES
    assert create_queries_and_maps(cats, own) == create_queries_and_maps(cats, hf)


def test_wordpiece_tokenizer_fuzz(tmp_path):
    from hypothesis import given, settings, strategies as st
    from transformers import BertTokenizerFast
    from hipie_amd.tokenizer import BertWordPiece
    vf = _wordpiece_vocab(tmp_path)
    own, hf = BertWordPiece(vf), BertTokenizerFast(vocab=vf, do_lower_case=True)
    alphabet = "abcdeghilnoprstuy ABCXZ.,-()'!?0129\t éïç中"

    @settings(max_examples=150, deadline=None)
    @given(st.text(alphabet=alphabet, max_size=40))
    def run(text):
        _check_tokenizer_equal(own, hf, text)
    run()


def test_test_time_mapper(tmp_path):
    """hipie_amd.mapper.TestTimeMapper: the is_train == False path of DetrDatasetMapperUni (coco_dataset_mapper_uni.py:453-600)."""
    import numpy as np
    from PIL import Image
    from hipie_amd.mapper import TestTimeMapper
    from hipie_amd.predictor import resize_shortest_edge_shape
    from hipie_amd.tokenizer import BertWordPiece
    tok = BertWordPiece(_wordpiece_vocab(tmp_path))
    labels = [{"id": 1, "name": "person,child"}, {"id": 2, "name": "invalid_class_id"}, {"id": 3, "name": "traffic light"},
              {"id": 4, "name": "sky-other-merged"}]
    m = TestTimeMapper(tok, min_size_test=64, max_size_test=100).register_dataset("toy_pan", labels, thing_class_ids=[0, 1])
    m.register_dataset("toy_inw", labels)
    rgb = np.random.RandomState(0).randint(0, 256, (60, 90, 3), dtype=np.uint8)
    path = os.path.join(str(tmp_path), "img.png")
    Image.fromarray(rgb).save(path)
    out = m({"file_name": path, "height": 60, "width": 90, "task": "detection", "dataset_name": "toy_pan", "annotations": [1, 2]})
    nh, nw = resize_shortest_edge_shape(60, 90, 64, 100)
    assert (nh, nw) == (64, 96) and out["image"].shape == (3, nh, nw) and out["image"].dtype == torch.uint8
    assert (out["height"], out["width"]) == (60, 90) and "annotations" not in out
    assert out["expressions"] == "person,child. traffic light. sky-other-merged"       # invalid_class_id dropped, names cleaned
    assert out["is_thing"] == {0: False, 1: True, 2: True, 3: False}
    assert [c["name"] for c in out["open_seg_labels"]] == ["person,child", "traffic light", "sky-other-merged"]
    pm = out["positive_map_label_to_token"]
    toks = tok.convert_ids_to_tokens(tok(out["expressions"]).input_ids[0])
    assert [toks[i] for i in pm[2]] == ["traffic", "light"] and sorted(pm) == [1, 2, 3]
    assert m({"file_name": path, "task": "detection", "dataset_name": "toy_inw"})["is_thing"] == {0: True, 1: True, 2: True, 3: True}
    g = m({"file_name": path, "task": "grounding", "expressions": "the person on the left"})
    assert g["expressions"] == "the person on the left" and g["is_thing"][1] is True and "positive_map_label_to_token" not in g
    # BGR format flips the channels of what was read; a wrong size in the dataset dict is refused
    mb = TestTimeMapper(tok, 60, 100, img_format="BGR")
    assert torch.equal(mb({"file_name": path, "task": "grounding", "expressions": "x"})["image"], torch.as_tensor(rgb[:, :, ::-1].copy()).permute(2, 0, 1))
    with pytest.raises(ValueError):
        m({"file_name": path, "height": 61, "width": 90, "task": "grounding", "expressions": "x"})


def test_selection_and_exact_attention_refuse_host_tensors():
    """no CPU fallback behind the product's own kernels: the two-stage selection and the exact small attention raise on host tensors."""
    from hipie_amd import ops
    from hipie_amd.modeling.transformer import _select_topk
    with pytest.raises(RuntimeError):
        _select_topk(torch.randn(2, 100), 10)
    q = torch.randn(1, 4, 2, 32)
    with pytest.raises(RuntimeError):
        ops.attn_f32(q, q, q, 1.0)
    with pytest.raises(RuntimeError):
        ops.ms_deform_attn_backward(torch.zeros(1, 4, 1, 2), torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 1, 1, 1, 1, 2),
                                    torch.ones(1, 1, 1, 1, 1), torch.zeros(1, 1, 2))


def test_effective_cores_respects_affinity_and_is_positive():
    import bench              # repo root is on sys.path (tests/conftest.py)
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))


def test_msda_im2col_step_is_validated_like_the_reference_op():
    """ms_deform_attn_cuda.cu:50-52 / :112-114: `batch % min(batch, im2col_step) == 0` or the op refuses the call.  The kernel here does not
    chunk the batch, but a call the reference would refuse must not silently succeed (checked before anything touches the device)."""
    import pytest
    import torch
    from hipie_amd import ops
    v = torch.zeros(6, 5, 8, 32)
    sh, ls = torch.tensor([[1, 5]]), torch.tensor([0])
    loc, w = torch.zeros(6, 3, 8, 1, 4, 2), torch.zeros(6, 3, 8, 1, 4)
    with pytest.raises(RuntimeError, match=r"batch\(6\) must divide im2col_step\(4\)"):
        ops.ms_deform_attn_forward(v, sh, ls, loc, w, 4)
    with pytest.raises(RuntimeError, match=r"must divide im2col_step"):
        ops.ms_deform_attn_backward(v, sh, ls, loc, w, torch.zeros(6, 3, 256), 4)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):      # a valid step (3 divides 6) gets as far as the device check
        ops.ms_deform_attn_forward(v, sh, ls, loc, w, 3)


def test_convt2x2_row_maps_reproduce_the_transposed_convolution():
    """ops.convt2x2_split's row maps (ConvTranspose2d(k = 2, s = 2) as one linear per tap, each row written to its place in the up-sampled
    channels-last map): emulated with torch matmuls on the CPU against F.conv_transpose2d."""
    import torch.nn.functional as F
    from hipie_amd import ops
    B, H, W, Cin, Cout = 2, 3, 5, 8, 16
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, Cin, H, W, generator=g)
    wt = torch.randn(Cin, Cout, 2, 2, generator=g)
    bias = torch.randn(Cout, generator=g)
    rows = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin)
    out = torch.zeros(B * 2 * H * 2 * W, Cout)
    maps = ops._shuffle_maps(B, H, W, torch.device("cpu"))
    assert len(maps) == 4 and all(m.dtype == torch.int32 and m.numel() == B * H * W for m in maps)
    assert torch.equal(torch.cat(maps).sort().values, torch.arange(B * 4 * H * W, dtype=torch.int32))     # every output row exactly once
    for t, m in enumerate(maps):
        i, j = divmod(t, 2)
        out[m.long()] = rows @ wt[:, :, i, j] + bias
    want = F.conv_transpose2d(x, wt, bias, stride=2)
    assert torch.allclose(out.view(B, 2 * H, 2 * W, Cout).permute(0, 3, 1, 2), want, atol=1e-5)


def test_training_operator_functions_have_no_host_path():
    """hipie_amd.training.functions (mask contraction, dynamic mask head): forward and backward run on libhipie_mi355.so only -- host
    tensors raise, nothing falls back to PyTorch or to the oracle."""
    from hipie_amd import ops
    from hipie_amd.training import functions
    e, f = torch.zeros(1, 2, 8), torch.zeros(1, 8, 2, 2)
    with pytest.raises(RuntimeError):
        functions.mask_einsum(e, f)
    with pytest.raises(RuntimeError):
        ops.mask_einsum_backward(e, f, torch.zeros(1, 2, 2, 2))
    feats, refs, params = torch.zeros(1, 8, 2, 2), torch.zeros(3, 2), torch.zeros(3, 169)
    with pytest.raises(RuntimeError):
        functions.dynamic_mask(feats, refs, params, 3, 8, 2)
    with pytest.raises(RuntimeError):
        ops.dynamic_mask_backward(feats, refs, params, torch.zeros(3, 4, 4), 3, stride=8, up=2)
    import inspect
    src = inspect.getsource(functions)
    assert "oracle" not in src.replace("oracle's", "")


def test_bench_pins_each_rank_to_its_own_cores():
    """bench.pin_host_threads (round 6, --gpus N readiness): rank r of N gets the r-th slice of the usable cores and sizes torch's intra-op
    pool to it; one rank keeps everything.  Run in a child process (it changes the affinity)."""
    import subprocess
    import sys
    code = ("import os, sys, json; sys.path.insert(0, %r); import bench, torch; "
            "all_ = sorted(os.sched_getaffinity(0)); r = bench.pin_host_threads(1, 2); mine = sorted(os.sched_getaffinity(0)); "
            "print(json.dumps([all_, mine, list(r), torch.get_num_threads()]))" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    import json
    all_, mine, r, nthreads = json.loads(out.stdout.strip().splitlines()[-1])
    if len(all_) >= 2:
        per = len(all_) // 2
        assert mine == all_[per:2 * per] and r[0] == per and nthreads == max(1, min(per, 8))
    else:
        assert mine == all_
