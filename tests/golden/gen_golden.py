#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Run in the authoring container only (needs /root/reference):

    python tests/golden/gen_golden.py [name ...]

Each fixture is a small .npz holding the reference outputs (+ a json ``meta`` with the
configuration and the state_dict manifest).  Weights and inputs are rebuilt on both sides from
seeds (tests/golden/_synth.py), so the fixtures stay small.  The reference is imported through
tests/golden/ref_shim.py; the only substituted arithmetic is MSDeformAttnFunction ->
ms_deform_attn_core_pytorch, the reference's own CPU formulation
(ops/functions/ms_deform_attn_func.py:43-63).
"""
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _synth  # noqa: E402
import ref_shim  # noqa: E402

ref = ref_shim.ref
torch.set_grad_enabled(False)


def NS(**kw):
    return types.SimpleNamespace(**kw)


def save(name, meta, _full=(), **arrays):
    """tensors with more than _synth.MAX_ELEMS elements are stored as a strided subsample of the flattened
    tensor (meta["subsampled"][key] = [step, full_shape]); tests compare with _synth.subsample()."""
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    meta = dict(meta)
    meta["subsampled"] = {}
    for k in list(out):
        if out[k].size > _synth.MAX_ELEMS and k not in _full and not k.startswith(tuple(_full) or ("\0",)):
            step = _synth.sub_step(out[k].size)
            meta["subsampled"][k] = [step, list(out[k].shape)]
            out[k] = out[k].reshape(-1)[::step].copy()
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("wrote %-28s %8.1f KB  %s" % (name + ".npz", os.path.getsize(path) / 1024, sorted(arrays)))


def man_json(man):
    return {k: list(v) for k, v in man.items()}


# ------------------------------------------------------------------------------ configs
TINY = dict(
    backbone="vit", vit_embed_dim=160, vit_depth=3, vit_heads=2, vit_window=14, vit_window_blocks=[0, 1],
    vit_img_size=1024, vit_patch=16, vit_pretrain_img_size=224, vit_mlp_ratio=4.0,
    hidden_dim=256, nheads=8, dim_feedforward=256, enc_layers=2, dec_layers=2, num_feature_levels=4,
    enc_n_points=4, dec_n_points=4, num_queries=40, num_bg_queries=10, num_vl_layers=1, vl_hidden_dim=2048,
    lang_dim=768, mask_stride=4, ctrl_layers=3,
    md_num_queries=30, md_dec_layers=3, md_enc_layers=2, md_dim_feedforward=256, md_enc_dim_feedforward=256,
    md_mask_dim=256, md_conv_dim=256,
    bert_layers=2, bert_hidden=768, bert_heads=12, bert_intermediate=3072, bert_vocab=30522, bert_max_pos=512,
    pixel_mean=[123.675, 116.280, 103.530], pixel_std=[58.395, 57.120, 57.375],
)


def hipie_cfg(c):
    """The yacs keys the imported reference modules read, as a SimpleNamespace tree
    (values: configs/eval/image_joint_r50_pan_maskdino_ade_test.yaml + hipie/config.py defaults)."""
    fuse = NS(CLAMP_MIN_FOR_UNDERFLOW=True, CLAMP_MAX_FOR_OVERFLOW=True, CLAMP_BERTATTN_MIN_FOR_UNDERFLOW=True,
              CLAMP_BERTATTN_MAX_FOR_OVERFLOW=True, STABLE_SOFTMAX_2D=False, CLAMP_DOT_PRODUCT=True)
    model = NS(
        USE_EARLY_FUSION=True, USE_ADDITIONAL_BERT=False, VL_FUSION_USE_CHECKPOINT=False,
        DECOUPLE_TGT=True, STILL_TGT_FOR_BOTH=True, USE_IOU_BRANCH=True, STILL_CLS_FOR_ENCODER=True,
        DEVICE="cpu",
        LANGUAGE_BACKBONE=NS(MODEL_TYPE="bert-base-uncased", MAX_QUERY_LEN=256, N_LAYERS=1, LANG_DIM=c["lang_dim"],
                             USE_CHECKPOINT=False),
        DYHEAD=NS(PRIOR_PROB=0.01, LOG_SCALE=0.0, FUSE_CONFIG=fuse),
        DDETRS=NS(HIDDEN_DIM=c["hidden_dim"], NUM_VL_LAYERS=c["num_vl_layers"], VL_HIDDEN_DIM=c["vl_hidden_dim"],
                  ENC_LAYERS=c["enc_layers"], TWO_STAGE_NUM_BG_PROPOSALS=c["num_bg_queries"],
                  TWO_STAGE_NUM_PROPOSALS=c["num_queries"], CTRL_LAYERS=c["ctrl_layers"]),
        PARALLEL_DET=False,
    )
    return NS(MODEL=model)


# ------------------------------------------------------------------------------ MSDA
def gen_msda():
    fn = ref("models.deformable_detr.ops.functions.ms_deform_attn_func")
    core = fn.ms_deform_attn_core_pytorch
    arrays, meta = {}, {"cases": []}
    # (1) the reference's own test recipe, ops/test.py:21-36,52-66 (CPU instead of .cuda())
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    for tag, dt in (("ref_double", torch.float64), ("ref_float", torch.float32)):
        value = torch.rand(N, S, M, D) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        attn = torch.rand(N, Lq, M, L, P) + 1e-5
        attn /= attn.sum(-1, keepdim=True).sum(-2, keepdim=True)
        out = core(value.to(dt), shapes, loc.to(dt), attn.to(dt))
        arrays.update({tag + "_value": value, tag + "_loc": loc, tag + "_attn": attn, tag + "_out": out,
                       tag + "_shapes": shapes})
        meta["cases"].append(tag)
    # (2) hot-path geometry M=8, D=32, L=4, P=4, locations incl. out-of-range
    for tag, B, Lq, shp, seed in (("hot_enc", 2, 340, [(16, 16), (8, 8), (4, 4), (2, 2)], 11),
                                  ("hot_dec", 1, 37, [(16, 16), (8, 8), (4, 4), (2, 2)], 12),
                                  ("hot_rect", 2, 50, [(12, 20), (6, 10), (3, 5), (2, 3)], 13)):
        shapes = torch.as_tensor(shp, dtype=torch.long)
        S = int(shapes.prod(1).sum())
        g = torch.Generator().manual_seed(seed)
        value = torch.randn(B, S, 8, 32, generator=g)
        loc = torch.rand(B, Lq, 8, 4, 4, 2, generator=g) * 1.2 - 0.1
        attn = torch.softmax(torch.randn(B, Lq, 8, 16, generator=g), -1).view(B, Lq, 8, 4, 4)
        out = core(value, shapes, loc, attn)
        arrays.update({tag + "_out": out, tag + "_shapes": shapes})
        meta["cases"].append(tag)
        meta[tag] = dict(B=B, Lq=Lq, seed=seed)
    save("msda", meta, **arrays)


def gen_msda_bwd():
    """gradients of the reference's differentiable formulation (ms_deform_attn_core_pytorch, ops/functions/ms_deform_attn_func.py:41-62:
    F.grid_sample + autograd) in double -- what ops/test.py's gradcheck holds ms_deform_attn_backward to."""
    fn = ref("models.deformable_detr.ops.functions.ms_deform_attn_func")
    core = fn.ms_deform_attn_core_pytorch
    arrays, meta = {}, {"cases": []}
    cases = [("recipe", D) for D in (30, 32, 64, 71, 1025)] + [("hot", None)]
    with torch.enable_grad():
        for tag, D in cases:
            value, shapes, loc, attn, gout = _synth.msda_bwd_inputs(tag, D)
            value.requires_grad_(True), loc.requires_grad_(True), attn.requires_grad_(True)
            out = core(value, shapes, loc, attn)
            gv, gl, ga = torch.autograd.grad(out, (value, loc, attn), gout)
            name = tag if D is None else "%s%d" % (tag, D)
            arrays.update({name + "_out": out.detach(), name + "_gvalue": gv, name + "_gloc": gl, name + "_gattn": ga})
            meta["cases"].append([name, tag, D])
    save("msda_bwd", meta, _full=tuple(arrays), **arrays)


# ------------------------------------------------------------------------------ ViT attention / backbone
def gen_vit_attn():
    vit = ref("backbone.vit")
    arrays, meta = {}, {"cases": {}}
    cases = {
        # name: (dim, heads, input_size for rel-pos table, (B,H,W) of x)
        "window14": (160, 2, (14, 14), (3, 14, 14)),
        "global16": (160, 2, (64, 64), (2, 16, 16)),     # table 127 -> interpolated to 31
        "global64": (80, 1, (64, 64), (1, 64, 64)),      # the real 64x64 geometry, 1 head
        "global_rect": (160, 2, (64, 64), (1, 12, 20)),  # non-square token grid
        "global84": (80, 1, (64, 64), (1, 84, 84)),      # 1344-pixel images (BASELINE configs[4]): table 127 -> 167 (utils.py:75-86)
        "global128": (80, 1, (64, 64), (1, 64, 128)),    # a 1024 x 2048 image (eval yamls: MAX_SIZE_TEST 2048): 64 x 128 tokens, tables 127 / 127 -> 255
    }
    for name, (dim, heads, insz, (B, H, W)) in cases.items():
        m = vit.Attention(dim, num_heads=heads, qkv_bias=True, use_rel_pos=True, rel_pos_zero_init=True,
                          input_size=insz).eval()
        man = _synth.load_synth(m, seed=21)
        x = _synth.synth_tensor("x_" + name, (B, H, W, dim), seed=22) * 8.0
        arrays[name + "_out"] = m(x)
        meta["cases"][name] = dict(dim=dim, heads=heads, input_size=list(insz), x_shape=[B, H, W, dim],
                                   manifest=man_json(man))
    save("vit_attn", meta, **arrays)


def build_ref_vit(c):
    vit = ref("backbone.vit")
    from functools import partial
    m = vit.ViT(img_size=c["vit_img_size"], patch_size=c["vit_patch"], in_chans=3, embed_dim=c["vit_embed_dim"],
                depth=c["vit_depth"], num_heads=c["vit_heads"], drop_path_rate=0.0, window_size=c["vit_window"],
                mlp_ratio=c["vit_mlp_ratio"], qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                window_block_indexes=c["vit_window_blocks"], residual_block_indexes=[], use_rel_pos=True,
                out_feature="last_feat", use_act_checkpoint=False,
                pretrain_img_size=c["vit_pretrain_img_size"]).eval()
    return m


def gen_vit_backbone():
    c = TINY
    m = build_ref_vit(c)
    man = _synth.load_synth(m, seed=31)
    x = _synth.synth_tensor("vit_in", (2, 3, 256, 224 + 32), seed=32)
    out = m(x)
    save("vit_backbone", dict(cfg=c, manifest=man_json(man), x_shape=list(x.shape)),
         res3=out["res3"], res4=out["res4"], res5=out["res5"])


# ------------------------------------------------------------------------------ VL fusion
def gen_bi_attn():
    fh = ref("models.deformable_detr.fuse_helper")
    cfg = hipie_cfg(TINY)
    arrays, meta = {}, {"cases": {}}
    for name, (B, Nv, L, npad, scale) in {"L20": (2, 340, 20, 3, 1.0), "L600_pad": (1, 340, 600, 300, 1.0),
                                          "clamp": (1, 64, 20, 5, 3000.0)}.items():
        m = fh.BiAttentionBlockForCheckpoint(v_dim=256, l_dim=768, embed_dim=2048, num_heads=8, dropout=0.1,
                                             drop_path=0.0, init_values=1.0 / 6, cfg=cfg).eval()
        man = _synth.load_synth(m, seed=41)
        v = _synth.synth_tensor("v_" + name, (B, Nv, 256), seed=42) * 16 * scale
        l = _synth.synth_tensor("l_" + name, (B, L, 768), seed=43) * 27 * scale
        mask = torch.ones(B, L, dtype=torch.long)
        mask[:, L - npad:] = 0
        if B > 1:
            mask[1, L - 2 * npad:] = 0
        ov, ol = m(v, l, attention_mask_l=mask)
        arrays.update({name + "_v": ov, name + "_l": ol, name + "_mask": mask})
        meta["cases"][name] = dict(B=B, Nv=Nv, L=L, scale=scale, manifest=man_json(man))
    save("bi_attn", meta, **arrays)


# ------------------------------------------------------------------------------ BERT wrapper
def build_ref_bert(c):
    bm = ref("models.deformable_detr.bert_model")
    from transformers import BertConfig, BertModel
    conf = BertConfig(vocab_size=c["bert_vocab"], hidden_size=c["bert_hidden"], num_hidden_layers=c["bert_layers"],
                      num_attention_heads=c["bert_heads"], intermediate_size=c["bert_intermediate"],
                      max_position_embeddings=c["bert_max_pos"], hidden_dropout_prob=0.0,
                      attention_probs_dropout_prob=0.0)
    enc = bm.BertEncoder.__new__(bm.BertEncoder)
    nn.Module.__init__(enc)
    enc.model = BertModel(conf, add_pooling_layer=False)
    enc.language_dim, enc.num_layers, enc.parallel_det = 768, 1, False
    return enc.eval()


def gen_bert():
    c = TINY
    enc = build_ref_bert(c)
    man = _synth.load_synth(enc, seed=51)
    arrays = {}
    ids, mask, _ = _synth.synth_token_ids(2, 9, 64, seed=52)
    arrays["short_hidden"] = enc({"input_ids": ids, "attention_mask": mask}, sep=1012)["hidden"]
    arrays["short_ids"], arrays["short_mask"] = ids, mask
    ids, mask, _ = _synth.synth_token_ids(2, 400, 1400, seed=53, pad_to=1536)
    out = enc({"input_ids": ids.clone(), "attention_mask": mask}, sep=1012)
    arrays["long_hidden"] = out["hidden"]
    arrays["long_ids"], arrays["long_mask"] = ids, mask
    save("bert", dict(cfg=c, manifest=man_json(man)), **arrays)


# ------------------------------------------------------------------------------ dynamic mask head (CondInst)
def gen_dynamic_mask():
    dn = ref("models.ddetrs_dn")
    fake = NS(no_rel_pos=False, dynamic_mask_channels=8, weight_nums=[80, 64, 8], bias_nums=[8, 8, 1],
              mask_out_stride=4, use_raft=False)
    fake.mask_heads_forward = types.MethodType(dn.DDETRSegmUniDN.mask_heads_forward, fake)
    arrays, meta = {}, {"cases": {}}
    for name, (B, Q, H, W) in {"sq": (2, 37, 32, 32), "rect": (1, 21, 20, 28)}.items():
        feats = _synth.synth_tensor("dm_feats_" + name, (B, 8, H, W), seed=61) * 20
        refs = torch.rand(1, B * Q, 2, generator=torch.Generator().manual_seed(62)) * torch.tensor([W * 8.0, H * 8.0])
        params = _synth.synth_tensor("dm_params_" + name, (1, B * Q, 169), seed=63) * 6
        out = dn.DDETRSegmUniDN.dynamic_mask_with_coords(fake, feats, refs, params, num_insts=[Q] * B,
                                                         mask_feat_stride=8, rel_coord=True, up_masks=None)
        arrays.update({name + "_out": out, name + "_refs": refs})
        meta["cases"][name] = dict(B=B, Q=Q, H=H, W=W)
    # aligned_bilinear on its own
    x = _synth.synth_tensor("ab_x", (3, 1, 9, 13), seed=64) * 10
    arrays["aligned_bilinear_out"] = dn.aligned_bilinear(x, 2)
    save("dynamic_mask", meta, **arrays)


# ------------------------------------------------------------------------------ full model (a22 parity surface)
class _ImageList:
    """the two things coco_inference needs from detectron2.structures.ImageList
    (hipie/models/ddetrs_dn.py:804-807): .image_sizes and iteration over unpadded images."""

    def __init__(self, tensors, image_sizes):
        self.tensor, self.image_sizes = tensors, image_sizes

    def __iter__(self):
        for t, (h, w) in zip(self.tensor, self.image_sizes):
            yield t[:, :h, :w]

    def __len__(self):
        return len(self.image_sizes)


def build_ref_maskdino(c, in_shape):
    enc_m = ref("models.maskdino.pixel_decoder.maskdino_encoder")
    dec_m = ref("models.maskdino.transformer_decoder.maskdino_decoder")
    head_m = ref("models.maskdino.meta_arch.maskdino_head")
    # values: configs/mask_dino/maskdino_R50_bs16_50ep_3s_dowsample1_2048.yaml
    pix = enc_m.MaskDINOEncoder(
        input_shape=in_shape, transformer_dropout=0.0, transformer_nheads=8,
        transformer_dim_feedforward=c["md_enc_dim_feedforward"], transformer_enc_layers=c["md_enc_layers"],
        conv_dim=c["md_conv_dim"], mask_dim=c["md_mask_dim"], norm="GN",
        transformer_in_features=["res3", "res4", "res5"], common_stride=4, num_feature_levels=3,
        total_num_feature_levels=4, feature_order="low2high")
    dec = dec_m.MaskDINODecoder(
        in_channels=c["md_conv_dim"], mask_classification=True, num_classes=c["hidden_dim"], hidden_dim=256,
        num_queries=c["md_num_queries"], nheads=8, dim_feedforward=c["md_dim_feedforward"],
        dec_layers=c["md_dec_layers"], mask_dim=c["md_mask_dim"], enforce_input_project=False, two_stage=True,
        dn="seg", noise_scale=0.4, dn_num=100, initialize_box_type="no", initial_pred=True, learn_tgt=False,
        total_num_feature_levels=4, dropout=0.0, semantic_ce_loss=False, dynamic_label_enc=True,
        dynamic_label_enc_dropout=0.1)
    head = head_m.MaskDINOHead(input_shape=in_shape, num_classes=c["hidden_dim"], pixel_decoder=pix,
                               loss_weight=1.0, ignore_value=255, transformer_predictor=dec)
    return head


def build_ref_model(c):
    """Assemble HIPIE_IMG.detr (DDETRSegmUniDN) from the reference's own classes, following
    hipie_img.py:77-176 and ddetrs_dn.py:90-215, without detectron2's config/registry layer."""
    cfg = hipie_cfg(c)
    mb_m = ref("backbone.masked_backbone")
    bb_m = ref("models.deformable_detr.backbone")
    pe_m = ref("models.deformable_detr.position_encoding")
    tr_m = ref("models.deformable_detr.deformable_transformer_dino")
    dd_m = ref("models.deformable_detr.deformable_detr")
    dn_m = ref("models.ddetrs_dn")
    if c["backbone"] == "r50":     # build_resnet_backbone (D2 resnet.py:614-694) with the R50 yaml's values
        rn = ref_shim.ref_d2_resnet()
        vit = rn.ResNet(rn.BasicStem(3, 64, norm="FrozenBN"), rn.ResNet.make_default_stages(50, norm="FrozenBN", stride_in_1x1=False),
                        out_features=["res3", "res4", "res5"])
        chans = [512, 1024, 2048]
    else:
        vit = build_ref_vit(c)
        E = c["vit_embed_dim"]
        chans = [E // 2, E, E]
    in_shape = {"res3": ref_shim.ShapeSpec(channels=chans[0], stride=8), "res4": ref_shim.ShapeSpec(channels=chans[1], stride=16),
                "res5": ref_shim.ShapeSpec(channels=chans[2], stride=32)}

    class _D2BB(vit.__class__):  # D2ViT.forward/output_shape/size_divisibility (vit.py:440-466)
        size_divisibility = 32

        def output_shape(self):
            return in_shape
    vit.__class__ = _D2BB
    masked = mb_m.MaskedBackbone.__new__(mb_m.MaskedBackbone)
    nn.Module.__init__(masked)
    masked.backbone = vit
    masked.feature_strides = [8, 16, 32]
    masked.num_channels = chans
    backbone = bb_m.Joiner(masked, pe_m.PositionEmbeddingSine(c["hidden_dim"] // 2, normalize=True))
    backbone.num_channels = masked.num_channels
    backbone.strides = masked.feature_strides
    transformer = tr_m.DeformableTransformerVLDINO(
        d_model=c["hidden_dim"], nhead=c["nheads"], num_encoder_layers=c["enc_layers"],
        num_decoder_layers=c["dec_layers"], dim_feedforward=c["dim_feedforward"], dropout=0.0, activation="relu",
        return_intermediate_dec=True, num_feature_levels=4, dec_n_points=4, enc_n_points=4, two_stage=True,
        two_stage_num_proposals=c["num_queries"], use_checkpoint=False, look_forward_twice=True,
        mixed_selection=True, cfg=cfg)
    detr = dd_m.DeformableDETRDINO(backbone, transformer, num_queries=c["num_queries"], num_feature_levels=4,
                                   aux_loss=True, with_box_refine=True, two_stage=True, mixed_selection=True, cfg=cfg)
    model = dn_m.DDETRSegmUniDN.__new__(dn_m.DDETRSegmUniDN)
    nn.Module.__init__(model)
    model.detr = detr
    model.rel_coord, model.ota, model.decouple_tgt, model.cls_pool_type = True, True, True, "average"
    model.use_iou_branch, model.new_mask_head, model.use_raft = True, False, False
    model.in_channels, model.dynamic_mask_channels, model.controller_layers = 8, 8, 3
    model.mask_out_stride, model.up_rate = 4, 2
    model.weight_nums, model.bias_nums, model.num_gen_params = [80, 64, 8], [8, 8, 1], 169
    model.controller = dn_m.MLP(256, 256, 169, 3)
    model.mask_head = dn_m.MaskHeadSmallConv(256, None, 256, use_raft=False, up_rate=2)
    model.resizer = tr_m.FeatureResizer(input_feat_size=768, output_feat_size=256, dropout=0.0)
    model.no_rel_pos, model.decouple_decoder = False, True
    model.mask_dino_fixed_linear_head, model.mask_dino_share_encoder, model.mask_dino_share_cls_head = False, False, False
    model.mask_dino = build_ref_maskdino(c, in_shape)
    model.feature_keys = ["res3", "res4", "res5"]
    model.mask_dino_cls_embed = dn_m._get_clones(detr.class_embed[0], c["md_dec_layers"] + 2)
    return model.eval()


def gen_refinit_stats():
    """tests/golden/refinit_stats.json: per-tensor statistics of the reference's OWN initialisation at the headline configuration -- the
    reference's classes are constructed (their constructors run _reset_parameters / _init_weights / the default nn init; the text encoder
    is transformers' BertModel(config)) and every state_dict entry is summarised as constant / normal / uniform / verbatim (see _synth.py).
    The second weight distribution of the parity study ("default module init (+N(0,0.02) for zero tensors)")."""
    torch.manual_seed(20260927)
    model = build_ref_model(FULL)
    bert = build_ref_bert(FULL)
    grid = _synth.msda_grid_bias()
    table = {}

    def summarise(prefix, module, canon):
        for k, v in module.state_dict().items():
            key = prefix + (canon(k) if canon else k)
            if not v.is_floating_point() or v.dim() == 0:
                continue
            x = v.detach().float().reshape(-1)
            if key in table:
                continue
            if k.endswith("sampling_offsets.bias") and x.numel() == grid.numel() and torch.allclose(x, grid, atol=1e-6):
                table[key] = ["g"]
            elif float(x.max()) == float(x.min()):
                table[key] = ["c", float(x[0])]
            elif x.numel() <= 64:
                table[key] = ["v", [float(t) for t in x]]
            else:
                mean, std = float(x.mean()), float(x.std())
                peak = float((x - mean).abs().max()) / std
                table[key] = ["u" if peak < 1.85 else "n", float("%.6g" % (mean if abs(mean) > 0.05 * std else 0.0)), float("%.6g" % std)]
    summarise("detr.", model, _synth.canonical_key)
    summarise("text_encoder.body.", bert, None)
    path = os.path.join(HERE, "refinit_stats.json")
    json.dump(table, open(path, "w"), indent=0, sort_keys=True)
    kinds = {}
    for v in table.values():
        kinds[v[0]] = kinds.get(v[0], 0) + 1
    print("wrote refinit_stats.json: %d tensors %s, %.1f KB" % (len(table), kinds, os.path.getsize(path) / 1024))
    # self-check: the table-driven draw has the statistics of the reference's draw, tensor by tensor
    sd = model.state_dict()
    dd = {k[len("detr."):]: v for k, v in table.items() if k.startswith("detr.")}
    worst = 0.0
    for k, v in sd.items():
        if not v.is_floating_point() or v.numel() < 4096:
            continue
        ck = _synth.canonical_key(k)
        mine = _synth.synth_tensor(k, v.shape, seed=71, dist=dd).float()
        if any(ck.endswith(d) for d in _synth._DEGENERATE) or (dd[ck][0] == "c" and float(mine.std()) > 0):
            continue                                 # the zero tensors that are deliberately redrawn
        a, b = float(v.float().std()), float(mine.std())
        if a > 0:
            worst = max(worst, abs(b / a - 1.0))
    print("refinit self-check: largest relative std mismatch over the big tensors %.3f" % worst)
    assert worst < 0.05


def gen_e2e_full_refinit():
    """e2e_full with the weights drawn from the reference's own initialisation distribution (refinit_stats.json)."""
    import time
    t0 = time.time()
    gen_e2e(FULL, "e2e_full_refinit", (("detection", 9),), sizes=((1024, 1024),), dist=_synth.refinit_stats())
    print("e2e_full_refinit: %.0f s" % (time.time() - t0))


def gen_e2e(c=None, name="e2e_tiny", tasks=(("detection", 9), ("grounding", 1)), sizes=((200, 256), (256, 224)), dist=None,
            imgs=None, prompts=None, extra_meta=None):
    """imgs / prompts: explicit inputs instead of the _synth ones -- prompts = {task: (ids (B, L), mask (B, L), pmap)} (gen_e2e_full_c80: the
    inputs bench.py times)."""
    c = c or TINY
    model = build_ref_model(c)
    bert = build_ref_bert(c)
    man = _synth.load_synth(model, seed=71, dist=dist, prefix="detr.")
    man_b = _synth.load_synth(bert, seed=72, dist=dist, prefix="text_encoder.body.")
    full_man = {"detr." + k: v for k, v in man.items()}
    full_man.update({"text_encoder.body." + k: v for k, v in man_b.items()})
    sizes = [tuple(s_) for s_ in sizes]
    imgs = _synth.synth_images(sizes, seed=73) if imgs is None else imgs
    mean = torch.tensor(c["pixel_mean"]).view(3, 1, 1)
    std = torch.tensor(c["pixel_std"]).view(3, 1, 1)
    # HIPIE_IMG.preprocess_image (hipie_img.py:880-898): normalise, ImageList.from_tensors (pad to max HxW)
    norm = [(x - mean) / std for x in imgs]
    Hm, Wm = max(s[0] for s in sizes), max(s[1] for s in sizes)
    batched = torch.zeros(len(imgs), 3, Hm, Wm)
    for i, x in enumerate(norm):
        batched[i, :, :x.shape[1], :x.shape[2]] = x
    images = _ImageList(batched, sizes)
    arrays, meta = {}, dict(cfg=c, manifest=man_json(full_man), sizes=sizes)
    if dist is not None:
        meta["dist"] = "refinit"
    meta.update(extra_meta or {})
    topk_log = []
    real_topk = torch.topk

    def spy_topk(*a, **k):
        r = real_topk(*a, **k)
        topk_log.append(r[1].clone())
        return r
    for spec in tasks:
        task, ncls = spec[0], spec[1]
        max_len, pad_to = (spec[2], spec[3]) if len(spec) > 2 else (64, None)
        if prompts is not None:
            ids, mask, pmap = prompts[task]
        else:
            ids, mask, pmap = _synth.synth_token_ids(2, ncls, max_len, seed=74, pad_to=pad_to)
        ids, mask = ids[:len(sizes)], mask[:len(sizes)]          # one prompt row per image (rows are generated independently)
        lang = bert({"input_ids": ids, "attention_mask": mask}, sep=1012)
        arrays[task + "_lang_hidden"] = lang["hidden"].clone()
        topk_log.clear()
        torch.topk = spy_topk
        try:
            out, _ = model.coco_inference(images, None, None, train=False,
                                          language_dict_features={"hidden": lang["hidden"].clone(), "masks": lang["masks"]},
                                          task=task, bg_queries_lang=None)
        finally:
            torch.topk = real_topk
        for k, v in out.items():
            if torch.is_tensor(v):
                arrays[task + "_" + k] = v
        arrays[task + "_topk_fg"] = topk_log[0]
        arrays[task + "_topk_md"] = topk_log[1]
        meta[task] = dict(n_classes=ncls, L=int(ids.shape[1]), max_len=max_len, pad_to=pad_to, pmap={str(k): v for k, v in pmap.items()})
    save(name, meta, **arrays)


def gen_e2e_r50():
    """the R50 configs (BASELINE configs[0]/[1]): same tiny heads behind the reference's detectron2 ResNet-50."""
    gen_e2e(dict(TINY, backbone="r50"), "e2e_r50_tiny", (("detection", 9), ("grounding", 1)))


def gen_e2e_long():
    """BASELINE configs[3]-style prompt inside the full path: ~815 real tokens (BertEncoder's > 512 chunking, bert_model.py:61-135),
    padded to 896 like PAD_MAX does; 400 synthetic classes."""
    gen_e2e(TINY, "e2e_long_tiny", (("detection", 400, 815, 896),))


DEEP = dict(TINY, vit_depth=32, vit_window_blocks=[0, 1, 3, 4, 6, 7, 9, 10], dim_feedforward=2048, enc_layers=6, dec_layers=6,
            num_queries=900, num_bg_queries=10, md_num_queries=300, md_dec_layers=9, md_enc_layers=6, md_dim_feedforward=2048,
            md_enc_dim_feedforward=2048, bert_layers=12)


def gen_e2e_deep():
    """the shipped DEPTHS on a narrow ViT: 32 blocks with the real window pattern (vit.py:412-421; dim 160 = 2 heads x 80 keeps
    the CPU run short), encoder 6 / decoder 6 / MaskDINO encoder 6 + decoder 9, FFN 2048, 900 + 10 / 300 queries, 12-layer
    BERT, two 256-pixel images -- rounding errors accumulate over the real number of layers (e2e_tiny is 3 / 2 / 2 / 3 deep)."""
    gen_e2e(DEEP, "e2e_deep", (("detection", 9),))


FULL = dict(DEEP, vit_embed_dim=1280, vit_heads=16)


def gen_e2e_full():
    """BASELINE.json's headline configuration itself: the full ViT-H (1280 wide, 16 heads, 32 blocks) and the shipped head sizes on ONE
    1024 x 1024 image with the 9-class prompt, through the reference's own coco_inference on the CPU (minutes).  Outputs larger than
    65536 elements are stored as strided subsamples (save())."""
    import time
    t0 = time.time()
    gen_e2e(FULL, "e2e_full", (("detection", 9),), sizes=((1024, 1024),))
    print("e2e_full: %.0f s" % (time.time() - t0))


def gen_e2e_full_c80():
    """The workload bench.py TIMES, literally (BASELINE configs[2]): the full ViT-H and the shipped head sizes on image 0 of
    bench.synth_batch (1024 x 1024) with the 80-class caption (L = 194 tokens), plus the separate grounding call of the same configuration
    (one referring expression, L = 12: tasks cannot mix, hipie_img.py:285), both through the reference's own coco_inference on the CPU.
    The test side rebuilds the inputs with bench.synth_batch (meta["bench_inputs"])."""
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import bench
    t0 = time.time()
    size, ncls, L = 1024, 80, 194
    det = bench.synth_batch(None, 1, size, ncls, L, "cpu", seed=0, task="detection")
    gnd = bench.synth_batch(None, 1, size, ncls, L, "cpu", seed=0, task="grounding")
    assert torch.equal(det[0]["image"], gnd[0]["image"])
    prompts = {"detection": (det[0]["input_ids"][None], det[0]["attention_mask"][None], det[0]["positive_map_label_to_token"]),
               "grounding": (gnd[0]["input_ids"][None], gnd[0]["attention_mask"][None], {1: list(range(1, 11))})}
    gen_e2e(FULL, "e2e_full_c80", (("detection", ncls), ("grounding", 1)), sizes=((size, size),), imgs=[det[0]["image"]], prompts=prompts,
            extra_meta={"bench_inputs": dict(size=size, n_classes=ncls, L=L, seed=0, batch=1)})
    print("e2e_full_c80: %.0f s" % (time.time() - t0))


def gen_e2e_r50_512():
    """BASELINE configs[0] literally: the reference's detectron2 ResNet-50 with the SHIPPED head sizes (encoder 6 / decoder 6 / MaskDINO
    encoder 6 + decoder 9, FFN 2048, 900 + 10 / 300 queries, 12-layer BERT) on ONE 512 x 512 image with ONE referring expression, through
    the reference's own coco_inference on the CPU (MODEL.DEVICE = cpu: about a minute); the detection task with the 9-class prompt rides along."""
    import time
    t0 = time.time()
    gen_e2e(dict(FULL, backbone="r50"), "e2e_r50_512", (("grounding", 1), ("detection", 9)), sizes=((512, 512),))
    print("e2e_r50_512: %.0f s" % (time.time() - t0))


def gen_e2e_padmax():
    """MODEL.LANGUAGE_BACKBONE.PAD_MAX with MAX_QUERY_LEN 4096, the shipped eval setting (configs/eval/image_joint_vit_huge_32g_pan_maskdino_ade_test.yaml:10-11,
    hipie_img.py:904-909): the e2e_tiny inputs with the 9-class caption padded to 4096 tokens.  BertEncoder's > 512 branch then leaves the
    hidden states of the padding zero (bert_model.py:118-127) and the fusion / class-logit stages run over 4096 mostly-masked columns."""
    gen_e2e(TINY, "e2e_padmax_tiny", (("detection", 9, 64, 4096),))


# ------------------------------------------------------------------------------ sub-module goldens from the e2e model
def gen_stages(c=None, name="stages_tiny", sizes=((200, 256), (256, 224))):
    """Intermediate tensors of the same tiny model (detection task), for stage-by-stage checks:
    backbone+pos, input_proj, encoder memory, two-stage selection, decoder hs, MaskDINO pixel decoder."""
    c = c or TINY
    model = build_ref_model(c)
    bert = build_ref_bert(c)
    _synth.load_synth(model, seed=71)
    _synth.load_synth(bert, seed=72)
    sizes = [tuple(s_) for s_ in sizes]
    imgs = _synth.synth_images(sizes, seed=73) if imgs is None else imgs
    mean = torch.tensor(c["pixel_mean"]).view(3, 1, 1)
    std = torch.tensor(c["pixel_std"]).view(3, 1, 1)
    batched = torch.zeros(len(sizes), 3, max(s_[0] for s_ in sizes), max(s_[1] for s_ in sizes))
    for i, x in enumerate(imgs):
        batched[i, :, :x.shape[1], :x.shape[2]] = (x - mean) / std
    misc = ref("util.misc")
    samples = misc.nested_tensor_from_tensor_list(list(_ImageList(batched, sizes)), size_divisibility=32)
    feats, pos = model.detr.backbone(samples)
    arrays = {}
    for i, (f, p) in enumerate(zip(feats, pos)):
        arrays["feat%d" % i], arrays["pos%d" % i], arrays["mask%d" % i] = f.tensors, p, f.mask
    ids, mask, _ = _synth.synth_token_ids(2, 9, 64, seed=74)
    ids, mask = ids[:len(sizes)], mask[:len(sizes)]
    lang = bert({"input_ids": ids, "attention_mask": mask}, sep=1012)
    caps = {}
    tr = model.detr.transformer

    def hook(name):
        def f(mod, inp, out):
            caps[name] = out
        return f
    hs = [tr.encoder.register_forward_hook(hook("encoder")),
          tr.encoder.vl_layers[0].register_forward_hook(hook("vl0")),
          tr.decoder.register_forward_hook(hook("decoder")),
          model.mask_head.register_forward_hook(hook("mask_head")),
          model.mask_dino.pixel_decoder.transformer.register_forward_hook(hook("md_enc"))]
    md_pix = {}
    orig_ff = model.mask_dino.pixel_decoder.forward_features

    def ff(features, masks):
        r = orig_ff(features, masks)
        md_pix["mask_features"], md_pix["ms"] = r[0], r[2]
        return r
    model.mask_dino.pixel_decoder.forward_features = ff
    out, _ = model.coco_inference(_ImageList(batched, sizes), None, None, train=False,
                                  language_dict_features={"hidden": lang["hidden"].clone(), "masks": lang["masks"]},
                                  task="detection", bg_queries_lang=None)
    for h in hs:
        h.remove()
    arrays["vl0_visual"] = caps["vl0"]["visual"]
    arrays["vl0_lang"] = caps["vl0"]["lang"]["hidden"]
    arrays["memory"] = caps["encoder"]["visual"]
    arrays["dec_hs"], arrays["dec_refs"] = caps["decoder"][0], caps["decoder"][1]
    arrays["mask_head_out"] = caps["mask_head"]
    arrays["md_enc_memory"] = caps["md_enc"][0]
    arrays["md_mask_features"] = md_pix["mask_features"]
    for i, t in enumerate(md_pix["ms"]):
        arrays["md_ms%d" % i] = t
    save(name, dict(cfg=c, sizes=sizes), **arrays)


def gen_stages_full():
    """the same intermediate tensors at BASELINE.json's headline configuration (full ViT-H, one 1024 x 1024 image; strided subsamples):
    where along the path an error of the product at full width comes from."""
    gen_stages(FULL, "stages_full", sizes=((1024, 1024),))


def gen_resnet50():
    """detectron2 ResNet-50 (the R50 configs' backbone): BasicStem + [3,4,6,3] BottleneckBlocks, FrozenBN, stride in the
    3x3 conv (STRIDE_IN_1X1 False), outputs res3/res4/res5 -- built from the reference's own resnet.py."""
    rn = ref_shim.ref_d2_resnet()
    stem = rn.BasicStem(3, 64, norm="FrozenBN")
    stages = rn.ResNet.make_default_stages(50, norm="FrozenBN", stride_in_1x1=False)
    m = rn.ResNet(stem, stages, out_features=["res3", "res4", "res5"]).eval()
    man = _synth.load_synth(m, seed=81)
    x = _synth.synth_tensor("r50_in", (1, 3, 64, 96), seed=82) * 2
    out = m(x)
    save("resnet50", dict(manifest=man_json(man), x_shape=list(x.shape)), res3=out["res3"], res4=out["res4"], res5=out["res5"])


# ------------------------------------------------------------------------------ post-processing
POST = dict(sizes=[(200, 256), (256, 224)], n_bg=10, n_fg=40, n_md=30, L=64, n_classes=9, seed=91,
            is_thing={"1": True, "2": True, "3": False, "4": True, "5": False, "6": False, "7": True, "8": True, "9": False})
POST_CASES = {
    # hipie/config.py defaults (:255-257)
    "default": dict(task="detection", use_bg_for_pano=True, bg_cls_agnostic=False, max_pool=False, out_hw=None),
    # configs/eval/image_joint_r50_pan_maskdino_ade_test.yaml:88-90 + evaluator-style output sizes
    "evalyaml": dict(task="detection", use_bg_for_pano=False, bg_cls_agnostic=True, max_pool=True, out_hw=[[150, 192], [300, 260]]),
    "grounding": dict(task="grounding", use_bg_for_pano=True, bg_cls_agnostic=False, max_pool=False, out_hw=None),
}


def gen_post():
    """HIPIE_IMG.inference (hipie_img.py:537-766) + panoptic_inference (:473-535) + semantic_inference (:870-878) +
    convert_grounding_to_od_logits (:1025-1052) + segmentation_postprocess (ddetrs.py:1029-1076), executed from the
    reference's own files on a synthetic a22 dictionary; CLIP fusion off (row f-2)."""
    m = ref_shim.ref_hipie_img()
    dd = ref("models.ddetrs")
    P = POST
    a22 = _synth.synth_a22([tuple(s) for s in P["sizes"]], P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"])
    _, _, pmap = _synth.synth_token_ids(2, P["n_classes"], P["L"], seed=74)
    is_thing = {int(k): v for k, v in P["is_thing"].items()}
    arrays, meta = {}, dict(post=P, cases=POST_CASES, pmap={str(k): v for k, v in pmap.items()})
    for cname, c in POST_CASES.items():
        me = NS(num_bg=P["n_bg"], num_fg=P["n_fg"], ota=True, mode_free_inference=False, max_pool_token_test=c["max_pool"],
                enable_clip=False, demo_only=False, mask_on=True, mask_stride=4, mask_thres=0.5,
                use_bg_for_pano=c["use_bg_for_pano"], bg_cls_agnostic=c["bg_cls_agnostic"], transform_eval=True, pano_temp=0.06,
                object_mask_threshold=0.25, overlap_threshold=0.8,
                detr=NS(bg_query_from_lang=False, decouple_decoder=True, mask_dino_fixed_linear_head=False))
        for fn in ("semantic_inference", "panoptic_inference"):
            setattr(me, fn, types.MethodType(getattr(m.HIPIE_IMG, fn), me))
        sizes = [tuple(s) for s in P["sizes"]]
        out_hw = [tuple(x) for x in c["out_hw"]] if c["out_hw"] else sizes
        grounding = c["task"] == "grounding"
        pm = {1: [0]} if grounding else pmap                                       # hipie_img.py:322-326
        out = {k: v.clone() for k, v in a22.items()}
        res = m.HIPIE_IMG.inference(me, out["pred_logits"], out["pred_boxes"], out["pred_masks"], sizes, pm, len(pm),
                                    task=c["task"], iou_pred=out["pred_boxious"], is_thing=[is_thing] * len(sizes),
                                    sizes=out_hw, output=out, bg_queries_lang=None, test_labels=None, images=None)
        for i, r in enumerate(res):
            inst = dd.segmentation_postprocess(r["instances"], out_hw[i][0], out_hw[i][1])   # hipie_img.py:358-362
            pre = "%s_%d_" % (cname, i)
            arrays[pre + "boxes"] = inst.pred_boxes.tensor
            arrays[pre + "scores"] = inst.scores
            arrays[pre + "classes"] = inst.pred_classes
            arrays[pre + "masks"] = np.packbits(inst.pred_masks.numpy().astype(bool), axis=None)
            meta[pre + "masks_shape"] = list(inst.pred_masks.shape)
            if not grounding:
                pan, info = r["panoptic_seg"]
                arrays[pre + "panoptic"] = pan.to(torch.int16)
                meta[pre + "segments"] = info
                arrays[pre + "semseg"] = r["sem_seg"]
    save("post", meta, _full=tuple(k for k in arrays if k.endswith(("masks", "panoptic"))), **arrays)


# ------------------------------------------------------------------------------ MaskCLIP score fusion
CLIP_TEST_LABELS = [{"name": "person,people"}, {"name": "dog"}, {"name": "traffic light"}, {"name": "sky,clouds"}, {"name": "zebra crossing"}]
CLIP_TRAIN_LABELS = [{"id": 1, "name": "person,child"}, {"id": 2, "name": "dog"}, {"id": 3, "name": "sky"}, {"id": 4, "name": "car"}]


def gen_maskclip():
    """the reference's OWN MaskCLIP (hipie/open_vocab/clip.py: get_mask_embed / encode_image_with_mask / _mask_clip_forward /
    pred_logits / build_text_embed) and HIPIE_IMG.get_clip_logits (hipie_img.py:811-868), plus HIPIE_IMG.inference with
    MODEL.CLIP.ENABLED on (:592-609, :735-747), executed over the open_clip stand-in of ref_shim (third-party architecture, tiny
    configuration, seeded weights).  One image: the reference's CLIP call asserts batch 1 (its evaluation batch size)."""
    clipm = ref_shim.ref_clip()
    m = ref_shim.ref_hipie_img()
    dd = ref("models.ddetrs")
    cfgc = ref_shim.OPEN_CLIP_CFG
    mc = clipm.MaskCLIP(name="tiny")
    man = _synth.load_synth(mc.clip, seed=95)
    arrays, meta = {}, dict(clip_cfg=cfgc, manifest=man_json(man), test_labels=CLIP_TEST_LABELS, train_labels=CLIP_TRAIN_LABELS)
    # (a) kernel level: mask embeddings, text embeddings, per-mask class logits, both fusion modes
    g = torch.Generator().manual_seed(96)
    image = torch.rand(1, 3, 70, 98, generator=g)
    ys, xs = torch.linspace(0, 1, 20).view(1, 1, 20, 1), torch.linspace(0, 1, 28).view(1, 1, 1, 28)
    cy, cx, r = torch.rand(1, 7, 1, 1, generator=g), torch.rand(1, 7, 1, 1, generator=g), 0.15 + 0.25 * torch.rand(1, 7, 1, 1, generator=g)
    mask = 6.0 * torch.tanh((r - torch.sqrt((ys - cy) ** 2 + (xs - cx) ** 2)) * 12.0) + 0.3 * torch.randn(1, 7, 20, 28, generator=g)
    mask[0, 6] = -8.0                                           # a mask that covers nothing: its token sees the class token only
    labels = clipm.prompt_labels([x["name"].split(",") for x in CLIP_TEST_LABELS], "photo") if hasattr(clipm, "prompt_labels") else \
        ref("open_vocab.helper").prompt_labels([x["name"].split(",") for x in CLIP_TEST_LABELS], "photo")
    text_embed = mc.build_text_embed(labels)
    out = mc(image, mask, text_embed, labels)
    arrays.update(image=image, mask=mask, text_embed=text_embed, mask_embed=out["mask_embed"], open_logits=out["mask_pred_open_logits"])
    prob = torch.softmax(torch.randn(7, len(CLIP_TEST_LABELS), generator=g), -1)
    arrays["pred_open_prob"] = prob
    me = NS(clip=mc, train_labels=CLIP_TRAIN_LABELS)
    me.get_clip_logits = types.MethodType(m.HIPIE_IMG.get_clip_logits, me)
    images = NS(tensor=image)
    for mode in ("MUL", "ADD"):
        me.clip_agg_mode = mode
        arrays["fused_" + mode] = me.get_clip_logits(0, [CLIP_TEST_LABELS], mask, images, prob, alpha=0.4, beta=0.45)
    # (b) HIPIE_IMG.inference with the fusion on: one image of the synthetic a22 dictionary, 5 classes
    P = dict(POST, sizes=[(200, 256)], n_classes=5, L=32, seed=97)
    a22 = _synth.synth_a22([tuple(s) for s in P["sizes"]], P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"])
    _, _, pmap = _synth.synth_token_ids(1, P["n_classes"], P["L"], seed=74)
    is_thing = {1: True, 2: True, 3: True, 4: False, 5: False}
    img255 = _synth.synth_images(P["sizes"], seed=98)[0]
    me = NS(num_bg=P["n_bg"], num_fg=P["n_fg"], ota=True, mode_free_inference=False, max_pool_token_test=False,
            enable_clip=True, clip=mc, train_labels=CLIP_TRAIN_LABELS, clip_alpha=0.4, clip_beta=0.45, clip_agg_mode="MUL", clip_fg_a=0.3,
            clip_fg_b=1.7, pano_temp_fg=0.06, demo_only=False, mask_on=True, mask_stride=4, mask_thres=0.5, use_bg_for_pano=True,
            bg_cls_agnostic=False, transform_eval=True, pano_temp=0.06, object_mask_threshold=0.25, overlap_threshold=0.8,
            detr=NS(bg_query_from_lang=False, decouple_decoder=True, mask_dino_fixed_linear_head=False))
    for fn in ("semantic_inference", "panoptic_inference", "get_clip_logits"):
        setattr(me, fn, types.MethodType(getattr(m.HIPIE_IMG, fn), me))
    sizes = [tuple(s) for s in P["sizes"]]
    out22 = {k: v.clone() for k, v in a22.items()}
    res = m.HIPIE_IMG.inference(me, out22["pred_logits"], out22["pred_boxes"], out22["pred_masks"], sizes, pmap, len(pmap),
                                task="detection", iou_pred=out22["pred_boxious"], is_thing=[is_thing], sizes=sizes, output=out22,
                                bg_queries_lang=None, test_labels=[CLIP_TEST_LABELS], images=NS(tensor=(img255 / 255.0)[None]))
    r = res[0]
    inst = dd.segmentation_postprocess(r["instances"], sizes[0][0], sizes[0][1])
    arrays["post_boxes"], arrays["post_scores"], arrays["post_classes"] = inst.pred_boxes.tensor, inst.scores, inst.pred_classes
    pan, info = r["panoptic_seg"]
    arrays["post_panoptic"] = pan.to(torch.int16)
    arrays["post_semseg"] = r["sem_seg"]
    meta.update(post=P, pmap={str(k): v for k, v in pmap.items()}, is_thing={str(k): v for k, v in is_thing.items()}, segments=info)
    save("maskclip", meta, _full=("post_panoptic", "image", "mask"), **arrays)


def gen_prompts():
    """create_queries_and_maps / create_positive_dict / clean_name of the reference's mapper
    (data/coco_dataset_mapper_uni.py:54-90, 732-736, 1024-1058).  That module cannot be imported (its package pulls in the
    dataset registry), so exactly those three function definitions are extracted from the reference file with ``ast`` and
    executed here, at generation time only; the fixture stores inputs and outputs."""
    import ast
    import re
    import tempfile
    from collections import defaultdict
    src = open(ref_shim.REF_HIPIE + "/data/coco_dataset_mapper_uni.py").read()
    tree = ast.parse(src)
    want = {"create_queries_and_maps", "create_positive_dict", "clean_name"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"re": re, "defaultdict": defaultdict}
    exec(compile(mod, "ref_mapper_functions", "exec"), ns)
    meta = {"categories": _synth.PROMPT_CATEGORIES, "vocab": _synth.PROMPT_VOCAB}
    with tempfile.TemporaryDirectory() as d:
        tok = _synth.prompt_tokenizer(d)
        for things_only in (False, True):
            q, pm = ns["create_queries_and_maps"](_synth.PROMPT_CATEGORIES, tok, things_only=things_only)
            meta["caption_%d" % things_only] = q
            meta["pmap_%d" % things_only] = {str(k): v for k, v in pm.items()}
    save("prompts", meta, dummy=np.zeros(1))


def gen_manifest_full():
    """state_dict names and shapes of the reference's full-size model (ViT-H backbone, the shipped head sizes): what a released
    checkpoint contains.  Built on the CPU from the reference's own classes (809.7 M parameters, ~30 s); only the
    name -> shape table is stored."""
    full = dict(TINY, vit_embed_dim=1280, vit_depth=32, vit_heads=16, vit_window_blocks=[0, 1, 3, 4, 6, 7, 9, 10],
                dim_feedforward=2048, enc_layers=6, dec_layers=6, num_queries=900, num_bg_queries=10, md_num_queries=300,
                md_dec_layers=9, md_enc_layers=6, md_dim_feedforward=2048, md_enc_dim_feedforward=2048, bert_layers=12)
    model, bert = build_ref_model(full), build_ref_bert(full)
    man = {"detr." + k: list(v.shape) for k, v in model.state_dict().items()}
    man.update({"text_encoder.body." + k: list(v.shape) for k, v in bert.state_dict().items()})
    with open(os.path.join(HERE, "manifest_vit_huge.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("wrote manifest_vit_huge.json  %d entries" % len(man))
    r50 = dict(full, backbone="r50")
    model = build_ref_model(r50)
    man = {"detr." + k: list(v.shape) for k, v in model.state_dict().items()}
    man.update({"text_encoder.body." + k: list(v.shape) for k, v in bert.state_dict().items()})
    with open(os.path.join(HERE, "manifest_r50.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print("wrote manifest_r50.json  %d entries" % len(man))


ALL = dict(manifest_full=gen_manifest_full, prompts=gen_prompts, post=gen_post, resnet50=gen_resnet50, msda=gen_msda, msda_bwd=gen_msda_bwd, vit_attn=gen_vit_attn, vit_backbone=gen_vit_backbone, bi_attn=gen_bi_attn, bert=gen_bert,
           dynamic_mask=gen_dynamic_mask, e2e=gen_e2e, stages=gen_stages, stages_full=gen_stages_full, e2e_r50=gen_e2e_r50, e2e_long=gen_e2e_long, e2e_deep=gen_e2e_deep, e2e_full=gen_e2e_full, maskclip=gen_maskclip, refinit_stats=gen_refinit_stats, e2e_full_refinit=gen_e2e_full_refinit, e2e_full_c80=gen_e2e_full_c80,
           e2e_padmax=gen_e2e_padmax, e2e_r50_512=gen_e2e_r50_512)

if __name__ == "__main__":
    names = sys.argv[1:] or list(ALL)
    for n in names:
        ALL[n]()
