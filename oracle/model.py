"""Model-level restatement of HIPIE_IMG's eval forward up to the a22 parity surface
(test infrastructure -- see oracle/__init__.py).

Functional fp32 PyTorch over a state_dict ``sd`` that uses the reference's key names
("detr.detr.transformer...", "text_encoder.body.model...").  ``cfg`` is a plain dict with
the hyper-parameters (the keys of hipie_amd.config.HipieConfig).
"""
import math

import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- small helpers
def lin(x, sd, p):
    return F.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], sd[p + "weight"], sd[p + "bias"], eps)


def mlp(x, sd, p, n):
    """MLP: Linear -> ReLU -> ... -> Linear (deformable_transformer_dino.py:599-633)."""
    for i in range(n):
        x = lin(x, sd, "%slayers.%d." % (p, i))
        if i < n - 1:
            x = F.relu(x)
    return x


def conv(x, sd, p, stride=1, padding=0):
    return F.conv2d(x, sd[p + "weight"], sd.get(p + "bias"), stride=stride, padding=padding)


def gn(x, sd, p, groups=32):
    return F.group_norm(x, groups, sd[p + "weight"], sd[p + "bias"], 1e-5)


def inverse_sigmoid(x, eps=1e-5):
    """hipie/util/misc.py:493-497."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# --------------------------------------------------------------------------- a1 preprocess
def preprocess(images, cfg):
    """HIPIE_IMG.preprocess_image (hipie_img.py:880-898) + nested_tensor_from_tensor_list(size_divisibility=32)
    (util/misc.py:288-316).  images: list of (3,h,w) float 0..255 -> tensors (B,3,Hp,Wp), mask (B,Hp,Wp) bool
    (True = padding), image_sizes."""
    mean = torch.tensor(cfg["pixel_mean"]).view(3, 1, 1)
    std = torch.tensor(cfg["pixel_std"]).view(3, 1, 1)
    sizes = [(int(x.shape[1]), int(x.shape[2])) for x in images]
    Hm = (max(s[0] for s in sizes) + 31) // 32 * 32
    Wm = (max(s[1] for s in sizes) + 31) // 32 * 32
    t = torch.zeros(len(images), 3, Hm, Wm)
    m = torch.ones(len(images), Hm, Wm, dtype=torch.bool)
    for i, x in enumerate(images):
        t[i, :, :sizes[i][0], :sizes[i][1]] = (x - mean) / std
        m[i, :sizes[i][0], :sizes[i][1]] = False
    return t, m, sizes


# --------------------------------------------------------------------------- a2 BERT
def bert_model(ids, mask, sd, p, cfg):
    """transformers.BertModel (third-party; bert_model.py:19,54-58 is the call site) restated from the
    published BERT-base algorithm: embeddings(word+position+type) -> LN(1e-12) -> N x [MHA(+mask) -> add&LN ->
    GELU(erf) FFN -> add&LN]; returns the last hidden state.  ``mask`` (B,L) of {0,1}: additive -finfo.min on
    masked keys (HF get_extended_attention_mask)."""
    B, L = ids.shape
    nh = cfg["bert_heads"]
    e = p + "embeddings."
    x = sd[e + "word_embeddings.weight"][ids] + sd[e + "position_embeddings.weight"][:L][None] \
        + sd[e + "token_type_embeddings.weight"][0][None, None]
    x = ln(x, sd, e + "LayerNorm.", 1e-12)
    ext = (1.0 - mask[:, None, None, :].float()) * torch.finfo(torch.float32).min
    for i in range(cfg["bert_layers"]):
        lp = "%sencoder.layer.%d." % (p, i)
        hd = x.shape[-1] // nh

        def heads(t):
            return t.view(B, L, nh, hd).permute(0, 2, 1, 3)
        q = heads(lin(x, sd, lp + "attention.self.query."))
        k = heads(lin(x, sd, lp + "attention.self.key."))
        v = heads(lin(x, sd, lp + "attention.self.value."))
        s = q @ k.transpose(-1, -2) / math.sqrt(hd) + ext
        ctx = (s.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, L, nh * hd)
        x = ln(lin(ctx, sd, lp + "attention.output.dense.") + x, sd, lp + "attention.output.LayerNorm.", 1e-12)
        h = F.gelu(lin(x, sd, lp + "intermediate.dense."))
        x = ln(lin(h, sd, lp + "output.dense.") + x, sd, lp + "output.LayerNorm.", 1e-12)
    return x


def bert_encoder(ids, mask, sd, p, cfg, sep=1012):
    """BertEncoder.forward (hipie/models/deformable_detr/bert_model.py:32-153), PARALLEL_DET False.
    <=512 tokens: one BertModel pass.  >512: split every sample at the last '.'/[SEP] before position 510,
    re-wrap chunks 2.. with [CLS] ... and a '.' after the chunk, batch the chunks, scatter the hidden states back.
    Returns {"masks": mask, "hidden": (B,L,768)}."""
    B, L = ids.shape
    if L <= 512:
        return {"masks": mask, "hidden": bert_model(ids, mask, sd, p, cfg)}
    CLS, EOS = 101, 102
    chunks = []
    for b in range(B):
        inp, msk = ids[b].clone(), mask[b]
        begin, start_src = 0, 0
        while True:
            seps = torch.where((inp == sep) | (inp == EOS))[0]
            seps = seps[seps < 510]
            if len(seps) == 0:
                break
            last = int(seps[-1])
            first = inp[:last + 1].clone()
            first[-1] = EOS
            fm = msk[begin:begin + last + 1] if False else None  # (kept for clarity; see below)
            # NOTE the reference slices the *original* mask row from 0 (mask_bs[:last_sep+1], :89) for every
            # chunk; valid tokens always form a prefix so chunk masks of later chunks are all ones as long as
            # the prefix of the row is valid -- reproduce that literally:
            first_mask = mask[b][:last + 1]
            on = torch.where(first_mask == 1)[0]
            n = len(first)
            out_mask = torch.zeros(512, dtype=ids.dtype)
            if start_src == 0:
                row = torch.cat([first, torch.zeros(512 - n, dtype=ids.dtype)])
                out_mask[on] = 1
            else:
                pad = torch.zeros(512 - n - 1, dtype=ids.dtype)
                pad[0] = sep
                row = torch.cat([torch.tensor([CLS], dtype=ids.dtype), first, pad])
                out_mask[on + 1] = 1
                out_mask[0] = 1
            chunks.append((b, row, out_mask, (start_src, start_src + n, begin, begin + n)))
            start_src = 1
            inp = inp[n:]
            begin += n
    rows = torch.stack([c[1] for c in chunks])
    masks = torch.stack([c[2] for c in chunks])
    hid = bert_model(rows, masks, sd, p, cfg)
    out = torch.zeros(B, L, hid.shape[-1])
    for i, (b, _, _, (s0, s1, t0, t1)) in enumerate(chunks):
        out[b, t0:t1] = hid[i, s0:s1]
    return {"masks": mask, "hidden": out}


# --------------------------------------------------------------------------- a3-a6 ViT backbone
def get_abs_pos(abs_pos, hw):
    """hipie/backbone/utils.py:128-157 with has_cls_token=True."""
    h, w = hw
    abs_pos = abs_pos[:, 1:]
    size = int(math.sqrt(abs_pos.shape[1]))
    if size != h or size != w:
        new = F.interpolate(abs_pos.reshape(1, size, size, -1).permute(0, 3, 1, 2), size=(h, w), mode="bicubic",
                            align_corners=False)
        return new.permute(0, 2, 3, 1)
    return abs_pos.reshape(1, h, w, -1)


def window_partition(x, ws):
    """hipie/backbone/utils.py:16-38."""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    """hipie/backbone/utils.py:41-60."""
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def vit_block(x, sd, p, heads, window):
    """Block.forward, hipie/backbone/vit.py:212-230 (LN eps 1e-6, exact-erf GELU MLP)."""
    shortcut = x
    x = ln(x, sd, p + "norm1.", 1e-6)
    if window > 0:
        H, W = x.shape[1], x.shape[2]
        x, pad_hw = window_partition(x, window)
    x = ops.vit_attention(x, sd, p + "attn.", heads)
    if window > 0:
        x = window_unpartition(x, window, pad_hw, (H, W))
    x = shortcut + x
    h = ln(x, sd, p + "norm2.", 1e-6)
    return x + lin(F.gelu(lin(h, sd, p + "mlp.fc1.")), sd, p + "mlp.fc2.")


def vit_backbone(x, sd, p, cfg):
    """ViT.forward, hipie/backbone/vit.py:357-374: patch conv -> +abs pos -> blocks -> {res3: ConvT x2,
    res4: identity, res5: maxpool /2}."""
    x = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=cfg["vit_patch"])
    x = x.permute(0, 2, 3, 1)
    x = x + get_abs_pos(sd[p + "pos_embed"], (x.shape[1], x.shape[2]))
    for i in range(cfg["vit_depth"]):
        win = cfg["vit_window"] if i in cfg["vit_window_blocks"] else 0
        x = vit_block(x, sd, "%sblocks.%d." % (p, i), cfg["vit_heads"], win)
    xp = x.permute(0, 3, 1, 2)
    return {"res3": F.conv_transpose2d(xp, sd[p + "fpn1.0.weight"], sd[p + "fpn1.0.bias"], stride=2),
            "res4": xp, "res5": F.max_pool2d(xp, 2, 2)}


# --------------------------------------------------------------------------- a7 masks + sine position
def pos_sine(mask, num_pos_feats=128, offset=-0.5):
    """PositionEmbeddingSine.forward (normalize=True, T=10000, scale 2pi, eps 1e-6).
    HIPIE branch: (embed - 0.5)/(last + eps) (deformable_detr/position_encoding.py:36-56) -> offset=-0.5;
    MaskDINO pixel decoder: embed/(last + eps) (maskdino/pixel_decoder/position_encoding.py:31-52) -> offset=0."""
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    y = (y + offset) / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = (x + offset) / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px = x[:, :, :, None] / dim_t
    py = y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def down_mask(m, size):
    """MaskedBackbone.forward: nearest resize of the pad mask (masked_backbone.py:21-29)."""
    return F.interpolate(m[None].float(), size=size).to(torch.bool)[0]


# --------------------------------------------------------------------------- a10 MSDeformAttn module
def msda_module(query, ref_points, src, shapes, pad_mask, sd, p, heads=8, levels=4, points=4):
    """MSDeformAttn.forward, ops/modules/ms_deform_attn.py:79-116."""
    N, Lq, C = query.shape
    S = src.shape[1]
    value = lin(src, sd, p + "value_proj.")
    if pad_mask is not None:
        value = value.masked_fill(pad_mask[..., None], 0.0)
    value = value.view(N, S, heads, C // heads)
    off = lin(query, sd, p + "sampling_offsets.").view(N, Lq, heads, levels, points, 2)
    aw = lin(query, sd, p + "attention_weights.").view(N, Lq, heads, levels * points)
    aw = F.softmax(aw, -1).view(N, Lq, heads, levels, points)
    shp = torch.as_tensor(shapes, dtype=torch.float32)
    if ref_points.shape[-1] == 2:
        norm = torch.stack([shp[:, 1], shp[:, 0]], -1)
        loc = ref_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
    else:
        loc = ref_points[:, :, None, :, None, :2] + off / points * ref_points[:, :, None, :, None, 2:] * 0.5
    out = ops.ms_deform_attn_core(value, shapes, loc, aw)
    return lin(out, sd, p + "output_proj.")


def encoder_ref_points(shapes, valid_ratios):
    """DeformableTransformerEncoderVL.get_reference_points, deformable_transformer_dino.py:313-325."""
    refs = []
    for lvl, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H), torch.linspace(0.5, W - 0.5, W), indexing="ij")
        ry = ry.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W)
        refs.append(torch.stack((rx, ry), -1))
    r = torch.cat(refs, 1)
    return r[:, :, None] * valid_ratios[:, None]


def valid_ratio(mask):
    """get_valid_ratio, deformable_transformer_dino.py:170-177."""
    _, H, W = mask.shape
    vh = torch.sum(~mask[:, :, 0], 1).float() / H
    vw = torch.sum(~mask[:, 0, :], 1).float() / W
    return torch.stack([vw, vh], -1)


def encoder_layer(src, pos, refs, shapes, pad_mask, sd, p):
    """DeformableTransformerEncoderLayer.forward, deformable_transformer_dino.py:384-394 (dropout 0)."""
    src2 = msda_module(src + pos, refs, src, shapes, pad_mask, sd, p + "self_attn.")
    src = ln(src + src2, sd, p + "norm1.")
    src2 = lin(F.relu(lin(src, sd, p + "linear1.")), sd, p + "linear2.")
    return ln(src + src2, sd, p + "norm2.")


def gen_proposals(memory, pad_mask, shapes):
    """gen_encoder_output_proposals (before enc_output), deformable_transformer_dino.py:138-166 and
    maskdino/utils/utils.py:33-71."""
    N = memory.shape[0]
    props, cur = [], 0
    for lvl, (H, W) in enumerate(shapes):
        m = pad_mask[:, cur:cur + H * W].view(N, H, W, 1)
        vH = torch.sum(~m[:, :, 0, 0], 1)
        vW = torch.sum(~m[:, 0, :, 0], 1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([vW.unsqueeze(-1), vH.unsqueeze(-1)], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        wh = torch.ones_like(grid) * 0.05 * (2.0 ** lvl)
        props.append(torch.cat((grid, wh), -1).view(N, -1, 4))
        cur += H * W
    prop = torch.cat(props, 1)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    prop = torch.log(prop / (1 - prop))
    prop = prop.masked_fill(pad_mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
    mem = memory.masked_fill(pad_mask.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0)
    return mem, prop


def sine_embed_4(pos, exchange_xy=True):
    """get_sine_pos_embed (deformable_transformer_dino.py:636-670) == gen_sineembed_for_position
    (maskdino/utils/utils.py:74-100): 128 feats per coordinate, T=10000, scale 2pi, output order (y,x,w,h)."""
    dim_t = torch.arange(128, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / 128)

    def f(x):
        s = x * (2 * math.pi) / dim_t
        return torch.stack((s[..., 0::2].sin(), s[..., 1::2].cos()), dim=-1).flatten(-2)
    res = [f(pos[..., i:i + 1]) for i in range(pos.shape[-1])]
    if exchange_xy:
        res[0], res[1] = res[1], res[0]
    return torch.cat(res, dim=-1)


def mha(x_qk, x_v, sd, p, heads=8):
    """nn.MultiheadAttention (q=k=tgt+pos, v=tgt, no mask, eval), batch-first here."""
    B, N, C = x_qk.shape
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    q = F.linear(x_qk, w[:C], b[:C])
    k = F.linear(x_qk, w[C:2 * C], b[C:2 * C])
    v = F.linear(x_v, w[2 * C:], b[2 * C:])
    hd = C // heads

    def sp(t):
        return t.view(B, N, heads, hd).transpose(1, 2)
    a = (sp(q) * hd ** -0.5) @ sp(k).transpose(-1, -2)
    o = (a.softmax(-1) @ sp(v)).transpose(1, 2).reshape(B, N, C)
    return lin(o, sd, p + "out_proj.")


def decoder_layer(tgt, qpos, refs_in, src, shapes, pad_mask, sd, p):
    """DeformableTransformerDecoderLayer.forward, deformable_transformer_dino.py:432-450 and
    maskdino/transformer_decoder/dino_decoder.py:221-270: self-attn -> norm2 -> MSDA cross -> norm1 -> FFN -> norm3."""
    qk = tgt + qpos
    tgt = ln(tgt + mha(qk, tgt, sd, p + "self_attn."), sd, p + "norm2.")
    t2 = msda_module(tgt + qpos, refs_in, src, shapes, pad_mask, sd, p + "cross_attn.")
    tgt = ln(tgt + t2, sd, p + "norm1.")
    t2 = lin(F.relu(lin(tgt, sd, p + "linear1.")), sd, p + "linear2.")
    return ln(tgt + t2, sd, p + "norm3.")


def vl_align(x, emb, sd, p):
    """VL_Align.forward, deformable_detr.py:55-73."""
    emb = F.normalize(emb, p=2, dim=-1)
    tok = lin(emb / 2.0, sd, p + "dot_product_projection_text.")
    bias = torch.matmul(emb, sd[p + "bias_lang"]) + sd[p + "bias0"]
    logit = torch.matmul(x, tok.transpose(-1, -2)) / sd[p + "log_scale"].exp() + bias.unsqueeze(1)
    return logit.clamp(max=50000).clamp(min=-50000)


def agg_lang_feat(hidden, mask):
    """deformable_transformer_dino.py:27-43, average pooling."""
    return (hidden * mask.unsqueeze(-1).float()).sum(1) / mask.sum(-1).unsqueeze(-1).float()


# --------------------------------------------------------------------------- HIPIE thing-branch transformer
def hipie_transformer(srcs, masks, poses, lang, sd, p, cfg, topk_override=None):
    """DeformableTransformerVLDINO.forward, deformable_transformer_dino.py:180-299 (eval, two-stage, DINO,
    DECOUPLE_TGT & STILL_TGT_FOR_BOTH).  Returns dict(hs, memory, init_ref, inter_refs, lang, topk, shapes)."""
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    mask = torch.cat([m.flatten(1) for m in masks], 1)
    pos = torch.cat([pe.flatten(2).transpose(1, 2) + sd[p + "level_embed"][i].view(1, 1, -1)
                     for i, pe in enumerate(poses)], 1)
    vr = torch.stack([valid_ratio(m) for m in masks], 1)
    refs = encoder_ref_points(shapes, vr)
    hidden, lmask = lang["hidden"], lang["masks"]
    for i in range(cfg["enc_layers"]):
        if i < cfg["num_vl_layers"]:
            src, hidden = ops.bi_attention_block(src, hidden, lmask, sd, "%sencoder.vl_layers.%d.b_attn." % (p, i))
        src = encoder_layer(src, pos, refs, shapes, mask, sd, "%sencoder.layers.%d." % (p, i))
    memory = src
    lang_pool = agg_lang_feat(hidden, lmask)
    # two-stage selection (:222-230)
    om, prop = gen_proposals(memory, mask, shapes)
    om = ln(lin(om, sd, p + "enc_output."), sd, p + "enc_output_norm.")
    nd = cfg["dec_layers"]
    enc_cls = lin(om, sd, "%sdecoder.class_embed.%d.body." % (p, nd))          # Still_Classifier
    enc_coord = mlp(om, sd, "%sdecoder.bbox_embed.%d." % (p, nd), 3) + prop
    topk = torch.topk(enc_cls[..., 0], cfg["num_queries"], dim=1)[1] if topk_override is None else topk_override
    ref = torch.gather(enc_coord, 1, topk.unsqueeze(-1).repeat(1, 1, 4)).sigmoid()
    bs = memory.shape[0]
    tgt = sd[p + "tgt_embed.weight"][None].repeat(bs, 1, 1)
    if cfg["num_bg_queries"] > 0:
        tgt = torch.cat([sd[p + "tgt_embed_bg.weight"][None].repeat(bs, 1, 1), tgt], 1)
        ref = torch.cat([sd[p + "bg_query_refs.weight"][None].repeat(bs, 1, 1), ref], 1)
    init_ref = ref
    # decoder (:467-525), look_forward_twice
    out, hs, inter = tgt, [], []
    for l in range(nd):
        ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
        qpos = mlp(sine_embed_4(ref_in[:, :, 0, :]), sd, p + "decoder.ref_point_head.", 2)
        out = decoder_layer(out, qpos, ref_in, memory, shapes, mask, sd, "%sdecoder.layers.%d." % (p, l))
        new_ref = (mlp(out, sd, "%sdecoder.bbox_embed.%d." % (p, l), 3) + inverse_sigmoid(ref)).sigmoid()
        ref = new_ref
        hs.append(out)
        inter.append(new_ref)
    return dict(hs=torch.stack(hs), memory=memory, init_ref=init_ref, inter_refs=torch.stack(inter),
                lang_hidden=hidden, topk=topk, shapes=shapes, enc_cls=enc_cls)


# --------------------------------------------------------------------------- MaskDINO branch (a20, a14, a21)
def maskdino_pixel_decoder(feats, sd, p, cfg):
    """MaskDINOEncoder.forward_features, maskdino/pixel_decoder/maskdino_encoder.py:368-434 with
    feature_order low2high, masks=None (all-False masks, valid_ratio 1 -- SURVEY 8a-1)."""
    f3, f4, f5 = feats["res3"], feats["res4"], feats["res5"]
    # input_proj index: 0..2 over reversed transformer_in_features = [res3,res4,res5]; 3 = extra stride-2 level on res5
    extra = gn(conv(f5, sd, p + "input_proj.3.0.", stride=2, padding=1), sd, p + "input_proj.3.1.")
    srcs = [gn(conv(f, sd, "%sinput_proj.%d.0." % (p, i)), sd, "%sinput_proj.%d.1." % (p, i))
            for i, f in enumerate((f3, f4, f5))] + [extra]
    B = f3.shape[0]
    zero = [torch.zeros(B, s.shape[2], s.shape[3], dtype=torch.bool) for s in srcs]
    poses = [pos_sine(z, 128, offset=0.0) for z in zero]   # pe_layer(x) depends only on the shape
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    t = p + "transformer."
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([pe.flatten(2).transpose(1, 2) + sd[t + "level_embed"][i].view(1, 1, -1)
                     for i, pe in enumerate(poses)], 1)
    mask = torch.cat([z.flatten(1) for z in zero], 1)
    vr = torch.ones(B, 4, 2)
    refs = encoder_ref_points(shapes, vr)
    for i in range(cfg["md_enc_layers"]):
        src = encoder_layer(src, pos, refs, shapes, mask, sd, "%sencoder.layers.%d." % (t, i))
    outs, st = [], 0
    for (H, W) in shapes:
        outs.append(src[:, st:st + H * W].transpose(1, 2).reshape(B, -1, H, W))
        st += H * W
    # one extra FPN level on res3 (num_fpn_levels = 1): lateral 1x1 (GN, no bias) + bilinear(out[0]) -> 3x3 GN ReLU
    cur = gn(F.conv2d(f3, sd[p + "adapter_1.weight"]), sd, p + "adapter_1.norm.")
    y = cur + F.interpolate(outs[0], size=cur.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(gn(F.conv2d(y, sd[p + "layer_1.weight"], padding=1), sd, p + "layer_1.norm."))
    mf = F.conv_transpose2d(y, sd[p + "mask_features.0.weight"], sd[p + "mask_features.0.bias"], stride=2)
    mf = F.relu(gn(mf, sd, p + "mask_features.1."))
    mf = conv(mf, sd, p + "mask_features.3.")
    return mf, outs, src


def maskdino_decoder(ms_feats, mask_features, sd, p, cfg, topk_override=None):
    """MaskDINODecoder.forward (eval), maskdino/transformer_decoder/maskdino_decoder.py:377-518 +
    TransformerDecoder.forward (dino_decoder.py:94-168) + forward_prediction_heads (:520-529) + pred_box (:357-375).
    Memory is flattened in REVERSED level order [s64,s32,s16,s8] (:385-397)."""
    nl = len(ms_feats)
    xs = [ms_feats[nl - 1 - i] for i in range(nl)]
    shapes = [tuple(x.shape[-2:]) for x in xs]
    src = torch.cat([x.flatten(2).transpose(1, 2) for x in xs], 1)      # input_proj is empty Sequential
    B = src.shape[0]
    mask = torch.zeros(B, src.shape[1], dtype=torch.bool)
    vr = torch.ones(B, nl, 2)
    om, prop = gen_proposals(src, mask, shapes)
    om = ln(lin(om, sd, p + "enc_output."), sd, p + "enc_output_norm.")
    cls_un = lin(om, sd, p + "class_embed.")
    coord_un = mlp(om, sd, p + "_bbox_embed.", 3) + prop
    nq = cfg["md_num_queries"]
    topk = torch.topk(cls_un.max(-1)[0], nq, dim=1)[1] if topk_override is None else topk_override
    ref_un = torch.gather(coord_un, 1, topk.unsqueeze(-1).repeat(1, 1, 4))
    tgt = torch.gather(om, 1, topk.unsqueeze(-1).repeat(1, 1, om.shape[-1]))

    def heads(x, pred_mask=True):
        d = ln(x, sd, p + "decoder_norm.")
        cls = lin(d, sd, p + "class_embed.")
        m = ops.mask_einsum(mlp(d, sd, p + "mask_embed.", 3), mask_features) if pred_mask else None
        return cls, m
    interm_cls, interm_mask = heads(tgt)
    ref = ref_un.sigmoid()
    refs, out, hs = [ref], tgt, []
    for l in range(cfg["md_dec_layers"]):
        ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
        qse = sine_embed_4(ref_in[:, :, 0, :])
        qpos = mlp(qse, sd, p + "decoder.ref_point_head.", 2)
        out = decoder_layer(out, qpos, ref_in, src, shapes, mask, sd, "%sdecoder.layers.%d." % (p, l))
        new_ref = (mlp(out, sd, p + "_bbox_embed.", 3) + inverse_sigmoid(ref)).sigmoid()
        ref = new_ref
        refs.append(new_ref)
        hs.append(ln(out, sd, p + "decoder.norm."))
    cls, m = heads(hs[-1])
    box = (mlp(hs[-1], sd, p + "_bbox_embed.", 3) + inverse_sigmoid(refs[-2])).sigmoid()
    return dict(pred_logits=cls, pred_masks=m, pred_boxes=box, topk=topk, interm_masks=interm_mask,
                interm_logits=interm_cls)


# --------------------------------------------------------------------------- CondInst mask branch (a17-a19)
def mask_head_small_conv(feats, sd, p):
    """MaskHeadSmallConv.forward with fpns=None, ddetrs_dn.py:1633-1689.  feats = [s8, s16, s32] NCHW."""
    x = F.relu(conv(feats[-1], sd, p + "lay3.", padding=1))
    x = feats[-2] + F.interpolate(x, size=feats[-2].shape[-2:], mode="nearest")
    x = F.relu(conv(x, sd, p + "lay4.", padding=1))
    x = feats[-3] + F.interpolate(x, size=feats[-3].shape[-2:], mode="nearest")
    x = F.relu(conv(x, sd, p + "jia_dcn.", padding=1))
    x = F.relu(conv(x, sd, p + "lay1.", padding=1))
    return F.relu(conv(x, sd, p + "lay2.", padding=1))


# --------------------------------------------------------------------------- a22 coco_inference
def coco_inference(images, lang, sd, cfg, task="detection", topk_fg=None, topk_md=None, want_stages=False):
    """HIPIE_IMG.forward (eval, hipie_img.py:314-337) -> DDETRSegmUniDN.coco_inference (ddetrs_dn.py:801-978).

    images: list of (3,h,w) float 0..255 RGB; lang: {"hidden": (B,L,768), "masks": (B,L)} from bert_encoder.
    topk_fg / topk_md pin the two discontinuous top-k selections to given indices (SURVEY 7 hard part (c)).
    Returns the a22 dict (+ "topk_fg", "topk_md", and stage tensors when want_stages)."""
    x, pad, sizes = preprocess(images, cfg)
    p = "detr.detr."
    if cfg.get("backbone", "vit") == "r50":
        feats = resnet50_backbone(x, sd, p + "backbone.0.backbone.")
    else:
        feats = vit_backbone(x, sd, p + "backbone.0.backbone.", cfg)
    names = ["res3", "res4", "res5"]
    fmasks = [down_mask(pad, feats[n].shape[-2:]) for n in names]
    poses = [pos_sine(m, cfg["hidden_dim"] // 2) for m in fmasks]
    srcs = [gn(conv(feats[n], sd, "%sinput_proj.%d.0." % (p, i)), sd, "%sinput_proj.%d.1." % (p, i))
            for i, n in enumerate(names)]
    s4 = gn(conv(feats["res5"], sd, p + "input_proj.3.0.", stride=2, padding=1), sd, p + "input_proj.3.1.")
    m4 = down_mask(fmasks[0], s4.shape[-2:])           # ddetrs_dn.py:841-842: resize of the LEVEL-0 mask
    srcs.append(s4)
    fmasks4 = fmasks + [m4]
    poses.append(pos_sine(m4, cfg["hidden_dim"] // 2))
    hidden0, lmask = lang["hidden"], lang["masks"]
    if task == "grounding":
        lang_pool0 = agg_lang_feat(hidden0, lmask).unsqueeze(1)
    tr = hipie_transformer(srcs, fmasks4, poses, {"hidden": hidden0, "masks": lmask}, sd, p + "transformer.", cfg,
                           topk_override=topk_fg)
    hs, inter = tr["hs"], tr["inter_refs"]
    # MaskDINO branch on the same backbone features, mask=None
    mf, ms, md_mem = maskdino_pixel_decoder(feats, sd, "detr.mask_dino.pixel_decoder.", cfg)
    md = maskdino_decoder(ms, mf, sd, "detr.mask_dino.predictor.", cfg, topk_override=topk_md)
    lang_for_md = lang_pool0 if task == "grounding" else tr["lang_hidden"]
    md_logits = vl_align(md["pred_logits"], lang_for_md, sd, "detr.mask_dino_cls_embed.%d." % (cfg["md_dec_layers"] + 1))
    # heads on the last decoder layer only (ddetrs_dn.py:898-933)
    lvl = cfg["dec_layers"] - 1
    reference = inverse_sigmoid(tr["init_ref"] if lvl == 0 else inter[lvl - 1])
    emb = lang_pool0 if task == "grounding" else tr["lang_hidden"]
    out = {}
    out["pred_logits"] = vl_align(hs[lvl], emb, sd, "%sclass_embed.%d." % (p, lvl))
    out["pred_boxes"] = (mlp(hs[lvl], sd, "%sbbox_embed.%d." % (p, lvl), 3) + reference).sigmoid()
    out["pred_boxious"] = lin(hs[lvl], sd, "%siou_head.%d." % (p, lvl))
    out["reference_points"] = inter[-2, :, :, :2]
    params = mlp(hs[lvl], sd, "detr.controller.", 3)
    B, nq, _ = params.shape
    refpts = torch.cat([out["reference_points"][i] * torch.tensor([float(w), float(h)])[None]
                        for i, (h, w) in enumerate(sizes)], 0)[None]
    mem, st, lv = tr["memory"], 0, []
    for (H, W) in tr["shapes"][:3]:
        lv.append(mem[:, st:st + H * W].reshape(B, H, W, -1).permute(0, 3, 1, 2))
        st += H * W
    mh = mask_head_small_conv(lv, sd, "detr.mask_head.")
    masks = ops.dynamic_mask(mh, refpts, params.reshape(1, B * nq, -1), [nq] * B, stride=8, up=8 // cfg["mask_stride"])
    out["pred_masks"] = masks.reshape(B, nq, 1, masks.shape[-2], masks.shape[-1])
    out["pred_masks_maskdino"] = md["pred_masks"]
    out["pred_logits_maskdino"] = md_logits
    out["pred_boxes_maskdino"] = md["pred_boxes"]
    out["topk_fg"], out["topk_md"] = tr["topk"], md["topk"]
    if want_stages:
        out["_stages"] = dict(feats=feats, poses=poses, fmasks=fmasks4, memory=tr["memory"], hs=hs, inter=inter,
                              lang_hidden=tr["lang_hidden"], mask_head=mh, md_mask_features=mf, md_ms=ms,
                              md_enc_memory=md_mem)
    return out


# --------------------------------------------------------------------------- ResNet-50 backbone (R50 configs 1-2)
def frozen_bn(x, sd, p, eps=1e-5):
    """detectron2.layers.FrozenBatchNorm2d (D2/layers/batch_norm.py:44-65): F.batch_norm(training=False)."""
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, eps)


def resnet50_backbone(x, sd, p):
    """detectron2 ResNet-50 (D2/modeling/backbone/resnet.py:330-359 BasicStem, :100-212 BottleneckBlock, :362-470 ResNet) with
    FrozenBN and STRIDE_IN_1X1 False (stride on the 3x3 conv): 7x7/2 conv + 3x3/2 max-pool, stages [3,4,6,3];
    returns res3 (512, /8), res4 (1024, /16), res5 (2048, /32)."""
    x = F.relu(frozen_bn(F.conv2d(x, sd[p + "stem.conv1.weight"], stride=2, padding=3), sd, p + "stem.conv1.norm."))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    out = {}
    for si, (name, nblk) in enumerate((("res2", 3), ("res3", 4), ("res4", 6), ("res5", 3))):
        for b in range(nblk):
            q = "%s%s.%d." % (p, name, b)
            stride = 2 if (b == 0 and si > 0) else 1
            y = F.relu(frozen_bn(F.conv2d(x, sd[q + "conv1.weight"]), sd, q + "conv1.norm."))
            y = F.relu(frozen_bn(F.conv2d(y, sd[q + "conv2.weight"], stride=stride, padding=1), sd, q + "conv2.norm."))
            y = frozen_bn(F.conv2d(y, sd[q + "conv3.weight"]), sd, q + "conv3.norm.")
            sc = x
            if (q + "shortcut.weight") in sd:
                sc = frozen_bn(F.conv2d(x, sd[q + "shortcut.weight"], stride=stride), sd, q + "shortcut.norm.")
            x = F.relu(y + sc)
        if name != "res2":
            out[name] = x
    return out
