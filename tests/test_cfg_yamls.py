"""SURVEY 8b1 on the REAL configuration surface: the 11 shipped eval yamls of the reference, merged by the reference's own config code.

* build container (needs /root/reference; skipped elsewhere): `add_hipie_config` (projects/HIPIE/hipie/config.py:5-284) over detectron2's own
  defaults.py, each configs/eval/*.yaml with its `_BASE_` inheritance, the MaskDINO yaml as hipie/models/maskdino/build.py:8-19 assembles
  it -- through the CfgNode stand-in of tests/golden/cfg_shim.py (yacs / fvcore are not installed; PyYAML is) -- into
  HipieConfig.from_yacs and a construction of HIPIE_IMG on the meta device; the flattened trees must equal tests/golden/eval_cfgs.json.
* everywhere (the GPU box too): the same constructions from the committed tables alone.
"""
import json
import os
import types

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
TABLES = os.path.join(HERE, "golden", "eval_cfgs.json")
HAVE_REF = os.path.isdir("/root/reference/projects/HIPIE/configs/eval")

# value -> yaml names (without the common prefix / suffix), read off the reference's yamls (configs/eval/*.yaml)
MAX_QUERY_LEN = {"r50_pan_maskdino_ade_test": 8192, "vit_huge_32g_pan_maskdino_ade_test": 4096, "r50_pan_maskdino_pascal": 1536,
                 "r50_pan_maskdino_voc": 1536, "vit_huge_32g_pan_maskdino_pascal": 1536, "vit_huge_32g_pan_maskdino_voc": 1536}


def _ns(flat):
    root = {}
    for k, v in flat.items():
        node = root
        parts = k.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = v

    def conv(d):
        return types.SimpleNamespace(**{k: conv(v) if isinstance(v, dict) else v for k, v in d.items()})
    return conv(root)


def _check_and_construct(name, cfg, md):
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    h = HipieConfig.from_yacs(cfg, md)
    short = name[len("image_joint_"):-len(".yaml")]
    assert h.backbone == ("vit" if "vit_huge" in short else "r50")
    if h.backbone == "vit":
        assert (h.vit_embed_dim, h.vit_depth, h.vit_heads) == (1280, 32, 16)
    assert h.max_query_len == MAX_QUERY_LEN.get(short, 1024) and h.pad_max is True          # every eval yaml pads to MAX_QUERY_LEN
    assert h.clip_enabled == (short != "r50_pan_maskdino_parts") and h.clip_name == "ViT-L-14-336" and h.clip_agg_mode == "MUL"
    assert 0.0 < h.clip_alpha <= 0.4 and 0.4 <= h.clip_beta <= 0.7 and (h.clip_fg_a, h.clip_fg_b) == (0.3, 1.7)
    assert (h.use_bg_for_pano, h.max_pool) == (False, True) and h.bg_cls_agnostic == (short != "r50_pan_maskdino_parts")
    assert (h.num_queries, h.num_bg_queries, h.md_num_queries, h.md_dec_layers, h.md_enc_layers) == (900, 10, 300, 9, 6)
    assert (h.enc_layers, h.dec_layers, h.dim_feedforward, h.md_dim_feedforward, h.md_enc_dim_feedforward) == (6, 6, 2048, 2048, 2048)
    assert h.pixel_mean == [123.675, 116.28, 103.53] and h.ota and h.object_mask_threshold == 0.25 and h.overlap_threshold == 0.8
    with torch.device("meta"):
        model = HIPIE_IMG(h, Precision.split3(), device="meta")
    # the state_dict a reference checkpoint of this configuration must fill: tests/golden/manifest_{vit_huge,r50}.json are the key / shape
    # lists of the REFERENCE's modules at these sizes (gen_golden.py manifest_full); the CLIP towers are not part of a HIPIE checkpoint
    man = json.load(open(os.path.join(HERE, "golden", "manifest_vit_huge.json" if h.backbone == "vit" else "manifest_r50.json")))
    own = {k: list(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip.")}
    assert own == {k: list(v) for k, v in man.items()}, sorted(set(own) ^ set(man))[:10]
    return h


@pytest.mark.skipif(not HAVE_REF, reason="/root/reference is not present (GPU box): the committed tables are checked instead")
def test_the_eleven_eval_yamls_through_the_references_own_config_code():
    import cfg_shim
    yamls = cfg_shim.eval_yamls()
    assert len(yamls) == 11
    committed = json.load(open(TABLES))
    keep = ("MODEL.", "TEST.", "INPUT.", "DATASETS.TEST", "VERSION")
    for y in yamls:
        cfg = cfg_shim.hipie_cfg(y)                                   # KeyError here = a yaml key add_hipie_config does not define
        md = cfg_shim.maskdino_cfg(cfg.MODEL.MASKDINO.CONFIG_PATH)
        name = os.path.basename(y)
        flat = {k: v for k, v in cfg_shim.flatten(cfg).items() if k.startswith(keep)}
        assert json.loads(json.dumps(flat)) == committed["eval"][name], name + ": tests/golden/eval_cfgs.json is stale (gen_cfg_golden.py)"
        _check_and_construct(name, cfg, md)


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference's config code")
def test_config_stand_in_has_yacs_semantics_and_unsupported_switches_raise():
    import cfg_shim
    y = [p for p in cfg_shim.eval_yamls() if p.endswith("vit_huge_32g_pan_maskdino_ade_test.yaml")][0]
    with pytest.raises(KeyError, match="Non-existent config key"):
        cfg_shim.hipie_cfg(y, opts=["MODEL.NO_SUCH_KEY", 1])
    with pytest.raises(ValueError, match="Type mismatch"):
        cfg_shim.hipie_cfg(y, opts=["MODEL.DDETRS.HIDDEN_DIM", "wide"])
    from hipie_amd.config import HipieConfig
    for key, val in (("MODEL.PARALLEL_DET", True), ("MODEL.DECOUPLE_TGT", False), ("MODEL.DDETRS.USE_DINO", False),
                     ("MODEL.MASKDINO.SHARE_CLS_HEAD", True), ("MODEL.MASKDINO.ENABLED", False)):
        cfg = cfg_shim.hipie_cfg(y, opts=[key, val])
        with pytest.raises(NotImplementedError, match=key.replace(".", r"\.")):
            HipieConfig.from_yacs(cfg, cfg_shim.maskdino_cfg(cfg.MODEL.MASKDINO.CONFIG_PATH))
    cfg = cfg_shim.hipie_cfg(y, opts=["MODEL.LANGUAGE_BACKBONE.MAX_QUERY_LEN", 256, "MODEL.LANGUAGE_BACKBONE.PAD_MAX", False])
    h = HipieConfig.from_yacs(cfg, cfg_shim.maskdino_cfg(cfg.MODEL.MASKDINO.CONFIG_PATH))
    assert (h.max_query_len, h.pad_max) == (256, False)


def test_the_eleven_eval_configurations_construct_from_the_committed_tables():
    t = json.load(open(TABLES))
    assert len(t["eval"]) == 11
    for name, flat in sorted(t["eval"].items()):
        md = _ns(t["maskdino"][flat["MODEL.MASKDINO.CONFIG_PATH"]])
        _check_and_construct(name, _ns(flat), md)
