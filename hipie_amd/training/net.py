"""The network of the TRAINING step: a differentiable forward over the parameters of the product model (the same
state_dict layout as the reference, hipie_amd/hipie_img.py), written for autograd instead of for speed -- the inference path's fused
kernels have no backward.  What is hand-written HIP here is what has a backward kernel: multi-scale deformable attention
(hipie_msda_forward / hipie_msda_backward), the mask contraction and the CondInst dynamic mask head (training/functions.py); the dense
linears, LayerNorm / GroupNorm, the softmax attentions and the convolutions run on the library kernels PyTorch-ROCm dispatches to, with
torch.autograd providing their backward.

Functional style over a dict ``sd`` of LIVE parameters (model.named_parameters() + buffers, reference key names), so gradients land in the
model's own parameters.  Every function cites the reference code it follows.  Training-mode differences from the inference path:
de-noising queries and their attention masks in both decoders, per-layer outputs, reference points detached between decoder layers
(deformable_transformer_dino.py:505, dino_decoder.py:160), the encoder's proposal logits / boxes for the encoder loss.

The three operator kernels come from a ``Backend``; the default is the HIP library and refuses host tensors.  tests/ plug the oracle's
restatements in to check the host logic without a GPU."""
import math

import torch
import torch.nn.functional as F


_LEVELS = {}


def level_tensors(shapes, device, which="int"):
    """the pyramid's [(H, W)] as device tensors, built once per geometry: (spatial_shapes int64, level_start_index) or, which="wh", the
    (L, 2) float (W, H) normaliser.  A host list -> device tensor is a pageable copy, i.e. a host wait for everything queued on the stream:
    two of them per MSDeformAttn call kept the host from ever running ahead of the GPU (127 ms per forward of it sitting there, sampled)."""
    key = (tuple((int(h), int(w)) for h, w in (shapes.tolist() if torch.is_tensor(shapes) else shapes)), str(device))
    if key not in _LEVELS:
        ss = torch.as_tensor(key[0], dtype=torch.int64, device=device)
        ls = torch.cat((ss.new_zeros(1), ss.prod(1).cumsum(0)[:-1]))
        _LEVELS[key] = (ss, ls, torch.stack((ss[:, 1], ss[:, 0]), -1).float())
    ss, ls, wh = _LEVELS[key]
    return wh if which == "wh" else (ss, ls)


class HipBackend:
    """the operator kernels with a hand-written backward (libhipie_mi355.so); no host path"""

    _owners = {}

    @classmethod
    def linear(cls, x, sd, p):
        """the big linears (ViT qkv / proj / mlp, encoder FFN) on the split-fp16 GEMM, forward and backward (functions.SplitLinearFunction);
        the HL8 weight copies are cached on one small owner object per parameter name"""
        from .functions import split_linear
        owner = cls._owners.setdefault(p, type("_W", (), {})())
        return split_linear(x, sd[p + "weight"], sd.get(p + "bias"), owner, "w")

    @staticmethod
    def msda(value, shapes, loc, aw):
        """value (B,S,M,D), shapes [(H,W)], loc (B,Lq,M,L,P,2), aw (B,Lq,M,L,P) -> (B,Lq,M*D)"""
        from ..msda_shim import MSDeformAttnFunction
        ss, ls = level_tensors(shapes, value.device)
        return MSDeformAttnFunction.apply(value.contiguous(), ss, ls, loc.contiguous(), aw.contiguous(), 64)

    @staticmethod
    def fused_attention(qa, ka, v):
        """softmax(q' k'^T) v of a ViT block on the fused split-fp16 kernels when the shape is covered (head width 80, <= 224 operand columns;
        large token counts that are not a multiple of 128 are padded with masked keys), else None: the caller's materialised formulation
        runs (the 196-token windows, where it is the faster one)"""
        from .functions import fused_attention
        return fused_attention(qa, ka, v)

    @staticmethod
    def mask_einsum(mask_embed, mask_features):
        from .functions import mask_einsum
        return mask_einsum(mask_embed, mask_features)

    @staticmethod
    def dynamic_mask(mask_feats, ref_points, params, num_insts, stride, up):
        """mask_feats (B,8,H,W); ref_points (sum n_i, 2) pixels; params (sum n_i, 169); num_insts per image -> (sum n_i, up*H, up*W)"""
        from .functions import dynamic_mask
        outs, st = [], 0
        for b, n in enumerate(num_insts):
            if n:
                outs.append(dynamic_mask(mask_feats[b:b + 1], ref_points[st:st + n], params[st:st + n], n, stride, up).reshape(n, up * mask_feats.shape[2], -1))
            st += n
        if not outs:
            return mask_feats.new_zeros(0, up * mask_feats.shape[2], up * mask_feats.shape[3])
        return torch.cat(outs, 0)


# ------------------------------------------------------------------------------------------------ small helpers
def lin(x, sd, p):
    return F.linear(x, sd[p + "weight"], sd.get(p + "bias"))


def ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, x.shape[-1:], sd[p + "weight"], sd[p + "bias"], eps)


def mlp(x, sd, p, n):
    """MLP (deformable_transformer_dino.py:599-633): Linear -> ReLU -> ... -> Linear"""
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
    """util/misc.py:493-497"""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


# ------------------------------------------------------------------------------------------------ ViT backbone (backbone/vit.py, utils.py)
_RESIZE_OPS = {}


def _bicubic_operator(src, dst, device, dtype=torch.float32):
    """(dst, src) matrix of F.interpolate(mode="bicubic", align_corners=False) along one axis, read off the library itself (the identity
    pushed through it), so the border clamping and the A = -0.75 kernel are the library's own"""
    key = (src, dst, str(device), dtype)
    if key not in _RESIZE_OPS:
        eye = torch.eye(src, device=device, dtype=dtype).view(1, src, src, 1)
        _RESIZE_OPS[key] = F.interpolate(eye, size=(dst, 1), mode="bicubic", align_corners=False)[0, :, :, 0].t().contiguous()
    return _RESIZE_OPS[key]


def get_abs_pos(abs_pos, hw):
    """utils.py:128-157 (has_cls_token).  The bicubic resize is linear and separable: it is applied as two small matrix products
    (rows, then columns) with the library's own weights -- the same map as F.interpolate, whose BACKWARD kernel
    (upsample_bicubic2d_backward_out_frame) takes 92 ms per step for this one (1, 1280, 64, 64) tensor on gfx950; as products it is 0.1 ms."""
    h, w = hw
    abs_pos = abs_pos[:, 1:]
    size = int(math.sqrt(abs_pos.shape[1]))
    if size != h or size != w:
        grid = abs_pos.reshape(size, size, -1)
        ah, aw = _bicubic_operator(size, h, abs_pos.device, abs_pos.dtype), _bicubic_operator(size, w, abs_pos.device, abs_pos.dtype)
        rows = torch.matmul(ah, grid.reshape(size, -1)).reshape(h, size, -1)                  # (h, size, C)
        return torch.matmul(aw, rows.permute(1, 0, 2).reshape(size, -1)).reshape(w, h, -1).permute(1, 0, 2)[None]
    return abs_pos.reshape(1, h, w, -1)


def window_partition(x, ws):
    """utils.py:16-38"""
    B, H, W, C = x.shape
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws
    if ph or pw:
        x = F.pad(x, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    x = x.view(B, Hp // ws, ws, Wp // ws, ws, C)
    return x.permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws, ws, C), (Hp, Wp)


def window_unpartition(win, ws, pad_hw, hw):
    """utils.py:41-60"""
    Hp, Wp = pad_hw
    H, W = hw
    B = win.shape[0] // (Hp * Wp // ws // ws)
    x = win.view(B, Hp // ws, Wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, Hp, Wp, -1)
    return x[:, :H, :W, :].contiguous()


def get_rel_pos(q_size, k_size, rel_pos):
    """utils.py:63-93"""
    n = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != n:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=n, mode="linear")
        r = r.reshape(-1, n).permute(1, 0)
    else:
        r = rel_pos
    dev = rel_pos.device
    qc = torch.arange(q_size, device=dev)[:, None] * max(k_size / q_size, 1.0)
    kc = torch.arange(k_size, device=dev)[None, :] * max(q_size / k_size, 1.0)
    rel = (qc - kc) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def _blin(x, sd, p, be):
    """a linear that carries flops: the backend's split-GEMM form when it has one (HipBackend.linear), else F.linear"""
    f = getattr(be, "linear", None)
    return f(x, sd, p) if f is not None else lin(x, sd, p)


_KEY_AXES = {}
FOLD_REL_POS = True


def _key_axis_indicators(H, W, device, dtype):
    """(H*W, H + W) constant: row n = key (n // W, n % W) has a one in column n // W and a one in column H + n % W"""
    key = (H, W, str(device), dtype)
    if key not in _KEY_AXES:
        n = torch.arange(H * W, device=device)
        ind = torch.zeros(H * W, H + W, device=device, dtype=dtype)
        ind[n, torch.div(n, W, rounding_mode="floor")] = 1
        ind[n, H + n % W] = 1
        _KEY_AXES[key] = ind
    return _KEY_AXES[key]


def vit_attention(x, sd, p, heads, be=None):
    """Attention.forward (vit.py:67-83) + add_decomposed_rel_pos (utils.py:96-125): the bias is computed from the UNSCALED q.
    The decomposed bias rel_h[q, kh] + rel_w[q, kw] is a product as well -- [rel_h(q, :), rel_w(q, :)] . [onehot(kh), onehot(kw)] -- so it
    rides in the logits' GEMM as H + W extra columns of the operands (q' = [scale q, rel_h, rel_w], k' = [k, indicators]) instead of two
    broadcast additions over the (heads, HW, HW) logits forward and two reductions over their gradient backward (at 64 x 64 tokens: four
    passes over 2 GB per block)."""
    B, H, W, C = x.shape
    hd = C // heads
    qkv = _blin(x, sd, p + "qkv.", be).reshape(B, H * W, 3, heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * heads, H * W, -1).unbind(0)
    Rh, Rw = get_rel_pos(H, H, sd[p + "rel_pos_h"]), get_rel_pos(W, W, sd[p + "rel_pos_w"])
    rq = q.reshape(B * heads, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", rq, Rh).reshape(B * heads, H * W, H)
    rel_w = torch.einsum("bhwc,wkc->bhwk", rq, Rw).reshape(B * heads, H * W, W)
    if FOLD_REL_POS:
        qa = torch.cat((q * hd ** -0.5, rel_h, rel_w), -1)
        ka = torch.cat((k, _key_axis_indicators(H, W, k.device, k.dtype).expand(B * heads, -1, -1)), -1)
        fused = getattr(be, "fused_attention", None)
        if fused is not None:
            o = fused(qa, ka, v)                      # HipBackend: forward + backward without the (heads, HW, HW) tensors (csrc/attn_train.hip)
            if o is not None:
                o = o.view(B, heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
                return _blin(o, sd, p + "proj.", be)
        attn = qa @ ka.transpose(-2, -1)
    else:
        attn = (q * hd ** -0.5) @ k.transpose(-2, -1)
        attn = (attn.view(B * heads, H, W, H, W) + rel_h.view(B * heads, H, W, H)[:, :, :, :, None]
                + rel_w.view(B * heads, H, W, W)[:, :, :, None, :]).view(B * heads, H * W, H * W)
    o = attn.softmax(dim=-1) @ v
    o = o.view(B, heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return _blin(o, sd, p + "proj.", be)


def vit_backbone(x, sd, p, cfg, be=None):
    """ViT.forward (vit.py:357-374) + the simple feature pyramid of D2ViT (fpn1 = ConvTranspose, identity, max pool)"""
    x = F.conv2d(x, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=cfg["vit_patch"]).permute(0, 2, 3, 1)
    x = x + get_abs_pos(sd[p + "pos_embed"], (x.shape[1], x.shape[2]))
    for i in range(cfg["vit_depth"]):
        bp = "%sblocks.%d." % (p, i)
        win = cfg["vit_window"] if i in cfg["vit_window_blocks"] else 0
        h = ln(x, sd, bp + "norm1.", 1e-6)
        if win > 0:
            H, W = h.shape[1], h.shape[2]
            h, pad_hw = window_partition(h, win)
        h = vit_attention(h, sd, bp + "attn.", cfg["vit_heads"], be)
        if win > 0:
            h = window_unpartition(h, win, pad_hw, (H, W))
        x = x + h
        h = ln(x, sd, bp + "norm2.", 1e-6)
        x = x + _blin(F.gelu(_blin(h, sd, bp + "mlp.fc1.", be)), sd, bp + "mlp.fc2.", be)
    xp = x.permute(0, 3, 1, 2)
    return {"res3": F.conv_transpose2d(xp, sd[p + "fpn1.0.weight"], sd[p + "fpn1.0.bias"], stride=2), "res4": xp, "res5": F.max_pool2d(xp, 2, 2)}


# ------------------------------------------------------------------------------------------------ masks, positions
def pos_sine(mask, num_pos_feats=128, offset=-0.5):
    """PositionEmbeddingSine (normalize, T 1e4, scale 2 pi): deformable_detr/position_encoding.py:36-56 (offset -0.5) and
    maskdino/pixel_decoder/position_encoding.py:31-52 (offset 0)"""
    not_mask = ~mask
    y = not_mask.cumsum(1, dtype=torch.float32)
    x = not_mask.cumsum(2, dtype=torch.float32)
    y = (y + offset) / (y[:, -1:, :] + 1e-6) * (2 * math.pi)
    x = (x + offset) / (x[:, :, -1:] + 1e-6) * (2 * math.pi)
    dim_t = torch.arange(num_pos_feats, dtype=torch.float32, device=mask.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / num_pos_feats)
    px, py = x[:, :, :, None] / dim_t, y[:, :, :, None] / dim_t
    px = torch.stack((px[:, :, :, 0::2].sin(), px[:, :, :, 1::2].cos()), dim=4).flatten(3)
    py = torch.stack((py[:, :, :, 0::2].sin(), py[:, :, :, 1::2].cos()), dim=4).flatten(3)
    return torch.cat((py, px), dim=3).permute(0, 3, 1, 2)


def down_mask(m, size):
    """MaskedBackbone.forward: nearest resize of the padding mask (masked_backbone.py:21-29)"""
    return F.interpolate(m[None].float(), size=size).to(torch.bool)[0]


def valid_ratio(mask):
    """deformable_transformer_dino.py:170-177"""
    _, H, W = mask.shape
    return torch.stack([torch.sum(~mask[:, 0, :], 1).float() / W, torch.sum(~mask[:, :, 0], 1).float() / H], -1)


def encoder_ref_points(shapes, vr):
    """deformable_transformer_dino.py:313-325"""
    refs = []
    for lvl, (H, W) in enumerate(shapes):
        ry, rx = torch.meshgrid(torch.linspace(0.5, H - 0.5, H, device=vr.device), torch.linspace(0.5, W - 0.5, W, device=vr.device), indexing="ij")
        ry = ry.reshape(-1)[None] / (vr[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (vr[:, None, lvl, 0] * W)
        refs.append(torch.stack((rx, ry), -1))
    return torch.cat(refs, 1)[:, :, None] * vr[:, None]


def sine_embed_4(pos):
    """get_sine_pos_embed (deformable_transformer_dino.py:636-670) == gen_sineembed_for_position (maskdino/utils/utils.py:74-100)"""
    dim_t = torch.arange(128, dtype=torch.float32, device=pos.device)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / 128)

    def f(x):
        s = x * (2 * math.pi) / dim_t
        return torch.stack((s[..., 0::2].sin(), s[..., 1::2].cos()), dim=-1).flatten(-2)
    res = [f(pos[..., i:i + 1]) for i in range(pos.shape[-1])]
    res[0], res[1] = res[1], res[0]
    return torch.cat(res, dim=-1)


def agg_lang_feat(hidden, mask):
    """deformable_transformer_dino.py:27-43 (average)"""
    return (hidden * mask.unsqueeze(-1).float()).sum(1) / mask.sum(-1).unsqueeze(-1).float()


def vl_align(x, emb, sd, p):
    """VL_Align.forward (deformable_detr.py:55-73)"""
    emb = F.normalize(emb, p=2, dim=-1)
    tok = lin(emb / 2.0, sd, p + "dot_product_projection_text.")
    bias = torch.matmul(emb, sd[p + "bias_lang"]) + sd[p + "bias0"]
    logit = torch.matmul(x, tok.transpose(-1, -2)) / sd[p + "log_scale"].exp() + bias.unsqueeze(1)
    return logit.clamp(max=50000).clamp(min=-50000)


# ------------------------------------------------------------------------------------------------ deformable attention, encoder / decoder layers
def msda_module(query, ref_points, src, shapes, pad_mask, sd, p, be, heads=8, levels=4, points=4):
    """MSDeformAttn.forward (ops/modules/ms_deform_attn.py:79-116)"""
    N, Lq, C = query.shape
    S = src.shape[1]
    value = lin(src, sd, p + "value_proj.")
    if pad_mask is not None:
        value = value.masked_fill(pad_mask[..., None], 0.0)
    value = value.view(N, S, heads, C // heads)
    off = lin(query, sd, p + "sampling_offsets.").view(N, Lq, heads, levels, points, 2)
    aw = F.softmax(lin(query, sd, p + "attention_weights.").view(N, Lq, heads, levels * points), -1).view(N, Lq, heads, levels, points)
    if ref_points.shape[-1] == 2:
        loc = ref_points[:, :, None, :, None, :] + off / level_tensors(shapes, query.device, "wh")[None, None, None, :, None, :]
    else:
        loc = ref_points[:, :, None, :, None, :2] + off / points * ref_points[:, :, None, :, None, 2:] * 0.5
    return lin(be.msda(value, shapes, loc, aw), sd, p + "output_proj.")


def encoder_layer(src, pos, refs, shapes, pad_mask, sd, p, be):
    """DeformableTransformerEncoderLayer.forward (deformable_transformer_dino.py:384-394), dropout 0"""
    src = ln(src + msda_module(src + pos, refs, src, shapes, pad_mask, sd, p + "self_attn.", be), sd, p + "norm1.")
    return ln(src + _blin(F.relu(_blin(src, sd, p + "linear1.", be)), sd, p + "linear2.", be), sd, p + "norm2.")


def mha(x_qk, x_v, sd, p, attn_mask=None, heads=8):
    """nn.MultiheadAttention (q = k = tgt + pos, v = tgt), batch first; attn_mask (Nq, Nq) bool, True = blocked"""
    B, N, C = x_qk.shape
    w, b = sd[p + "in_proj_weight"], sd[p + "in_proj_bias"]
    hd = C // heads

    def sp(t):
        return t.view(B, N, heads, hd).transpose(1, 2)
    q, k, v = sp(F.linear(x_qk, w[:C], b[:C])), sp(F.linear(x_qk, w[C:2 * C], b[C:2 * C])), sp(F.linear(x_v, w[2 * C:], b[2 * C:]))
    a = (q * hd ** -0.5) @ k.transpose(-1, -2)
    if attn_mask is not None:
        a = a.masked_fill(attn_mask[None, None], float("-inf"))
    return lin((a.softmax(-1) @ v).transpose(1, 2).reshape(B, N, C), sd, p + "out_proj.")


def decoder_layer(tgt, qpos, refs_in, src, shapes, pad_mask, sd, p, be, attn_mask=None):
    """DeformableTransformerDecoderLayer.forward (deformable_transformer_dino.py:432-450; maskdino dino_decoder.py:221-270)"""
    tgt = ln(tgt + mha(tgt + qpos, tgt, sd, p + "self_attn.", attn_mask), sd, p + "norm2.")
    tgt = ln(tgt + msda_module(tgt + qpos, refs_in, src, shapes, pad_mask, sd, p + "cross_attn.", be), sd, p + "norm1.")
    return ln(tgt + lin(F.relu(lin(tgt, sd, p + "linear1.")), sd, p + "linear2."), sd, p + "norm3.")


def gen_proposals(memory, pad_mask, shapes):
    """gen_encoder_output_proposals before enc_output (deformable_transformer_dino.py:138-166; maskdino/utils/utils.py:33-71)"""
    N, dev = memory.shape[0], memory.device
    props, cur = [], 0
    for lvl, (H, W) in enumerate(shapes):
        m = pad_mask[:, cur:cur + H * W].view(N, H, W, 1)
        vH, vW = torch.sum(~m[:, :, 0, 0], 1), torch.sum(~m[:, 0, :, 0], 1)
        gy, gx = torch.meshgrid(torch.linspace(0, H - 1, H, device=dev), torch.linspace(0, W - 1, W, device=dev), indexing="ij")
        grid = torch.cat([gx.unsqueeze(-1), gy.unsqueeze(-1)], -1)
        scale = torch.cat([vW.unsqueeze(-1), vH.unsqueeze(-1)], 1).view(N, 1, 1, 2)
        grid = (grid.unsqueeze(0).expand(N, -1, -1, -1) + 0.5) / scale
        props.append(torch.cat((grid, torch.ones_like(grid) * 0.05 * (2.0 ** lvl)), -1).view(N, -1, 4))
        cur += H * W
    prop = torch.cat(props, 1)
    valid = ((prop > 0.01) & (prop < 0.99)).all(-1, keepdim=True)
    prop = torch.log(prop / (1 - prop)).masked_fill(pad_mask.unsqueeze(-1), float("inf")).masked_fill(~valid, float("inf"))
    return memory.masked_fill(pad_mask.unsqueeze(-1), 0.0).masked_fill(~valid, 0.0), prop


# ------------------------------------------------------------------------------------------------ vision-language fusion (fuse_helper.py)
def bi_attention_block(v, l, text_mask, sd, p, heads=8, dropout=0.0):
    """BiAttentionBlockForCheckpoint.forward (fuse_helper.py:170-179) + BiMultiHeadAttention.forward (:54-139).  dropout: the functional
    attention dropout of :111-112 (0.1 in vlfusion.py:81; 0 reproduces the deterministic fixture)"""
    v, l = ln(v, sd, p + "layer_norm_v."), ln(l, sd, p + "layer_norm_l.")
    a = p + "attn."
    B, Nv, _ = v.shape
    L = l.shape[1]
    E = sd[a + "v_proj.weight"].shape[0]
    hd = E // heads

    def split(t, n):
        return t.view(B, n, heads, hd).transpose(1, 2).reshape(B * heads, n, hd)
    q = split(lin(v, sd, a + "v_proj.") * hd ** -0.5, Nv)
    k = split(lin(l, sd, a + "l_proj."), L)
    vv, vl = split(lin(v, sd, a + "values_v_proj."), Nv), split(lin(l, sd, a + "values_l_proj."), L)
    w = torch.bmm(q, k.transpose(1, 2)).clamp(min=-50000).clamp(max=50000)
    wT = w.transpose(1, 2)
    wl = (wT - torch.max(wT, dim=-1, keepdim=True)[0]).clamp(min=-50000).clamp(max=50000).softmax(dim=-1)
    am = text_mask.to(torch.int64)[:, None, None, :].expand(B, 1, Nv, L)
    am = am.masked_fill(am == 0, int(-9e15))                  # int64 mask: 0 -> -9e15, 1 stays +1 (fuse_helper.py:97-108)
    wv = F.softmax((w.view(B, heads, Nv, L) + am).view(B * heads, Nv, L), dim=-1)
    if dropout > 0:
        wv, wl = F.dropout(wv, dropout, True), F.dropout(wl, dropout, True)
    ov = torch.bmm(wv, vl).view(B, heads, Nv, hd).transpose(1, 2).reshape(B, Nv, E)
    ol = torch.bmm(wl, vv).view(B, heads, L, hd).transpose(1, 2).reshape(B, L, E)
    return v + sd[p + "gamma_v"] * lin(ov, sd, a + "out_v_proj."), l + sd[p + "gamma_l"] * lin(ol, sd, a + "out_l_proj.")


# ------------------------------------------------------------------------------------------------ the thing branch's transformer
def hipie_transformer(srcs, masks, poses, lang, sd, p, cfg, be, query_label=None, query_bbox=None, attn_mask=None, fusion_dropout=0.0, topk_override=None):
    """DeformableTransformerVLDINO.forward (deformable_transformer_dino.py:180-299): two-stage, mixed selection, DECOUPLE_TGT +
    STILL_TGT_FOR_BOTH, look-forward-twice.  query_label (B, P, 256) / query_bbox (B, P, 4, un-sigmoided) = the de-noising part in front of
    the queries, attn_mask (Nq, Nq) its self-attention mask.  Returns hs (layers, B, Nq, 256), memory, init_ref, inter_refs, the
    proposals' class logits and un-activated boxes (for the encoder loss), the fused language features."""
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    mask = torch.cat([m.flatten(1) for m in masks], 1)
    pos = torch.cat([pe.flatten(2).transpose(1, 2) + sd[p + "level_embed"][i].view(1, 1, -1) for i, pe in enumerate(poses)], 1)
    vr = torch.stack([valid_ratio(m) for m in masks], 1)
    refs = encoder_ref_points(shapes, vr)
    hidden, lmask = lang["hidden"], lang["masks"]
    for i in range(cfg["enc_layers"]):
        if i < cfg["num_vl_layers"]:
            src, hidden = bi_attention_block(src, hidden, lmask, sd, "%sencoder.vl_layers.%d.b_attn." % (p, i), dropout=fusion_dropout)
        src = encoder_layer(src, pos, refs, shapes, mask, sd, "%sencoder.layers.%d." % (p, i), be)
    memory = src
    lang_pool = agg_lang_feat(hidden, lmask)
    ref_feat = ln(lin(lang_pool, sd, p + "resizer.fc."), sd, p + "resizer.layer_norm.", 1e-12).unsqueeze(1)       # FeatureResizer, dropout 0
    om, prop = gen_proposals(memory, mask, shapes)
    om = ln(lin(om, sd, p + "enc_output."), sd, p + "enc_output_norm.")
    nd = cfg["dec_layers"]
    enc_cls = lin(om, sd, "%sdecoder.class_embed.%d.body." % (p, nd))          # Still_Classifier (STILL_CLS_FOR_ENCODER)
    enc_coord = mlp(om, sd, "%sdecoder.bbox_embed.%d." % (p, nd), 3) + prop
    topk = torch.topk(enc_cls[..., 0], cfg["num_queries"], dim=1)[1] if topk_override is None else topk_override
    ref = torch.gather(enc_coord, 1, topk.unsqueeze(-1).repeat(1, 1, 4)).sigmoid()
    bs = memory.shape[0]
    tgt = sd[p + "tgt_embed.weight"][None].repeat(bs, 1, 1)
    if cfg["num_bg_queries"] > 0:
        tgt = torch.cat([sd[p + "tgt_embed_bg.weight"][None].repeat(bs, 1, 1), tgt], 1)
        ref = torch.cat([sd[p + "bg_query_refs.weight"][None].repeat(bs, 1, 1), ref], 1)
    if query_bbox is not None:
        ref = torch.cat([query_bbox.sigmoid(), ref], 1)
    init_ref = ref
    if query_label is not None:
        tgt = torch.cat([query_label, tgt], 1)
    out = tgt + 0.0 * ref_feat                                                  # decouple_tgt & still_tgt_for_both (:262-266)
    hs, inter = [], []
    for l in range(nd):
        ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
        qpos = mlp(sine_embed_4(ref_in[:, :, 0, :]), sd, p + "decoder.ref_point_head.", 2)
        out = decoder_layer(out, qpos, ref_in, memory, shapes, mask, sd, "%sdecoder.layers.%d." % (p, l), be, attn_mask)
        new_ref = (mlp(out, sd, "%sdecoder.bbox_embed.%d." % (p, l), 3) + inverse_sigmoid(ref)).sigmoid()
        ref = new_ref.detach()                                                  # :505
        hs.append(out)
        inter.append(new_ref)                                                   # look_forward_twice
    return dict(hs=torch.stack(hs), memory=memory, init_ref=init_ref, inter_refs=torch.stack(inter), lang_hidden=hidden, topk=topk, shapes=shapes,
                enc_cls=enc_cls, enc_coord=enc_coord, valid_ratios=vr)


# ------------------------------------------------------------------------------------------------ MaskDINO branch
def maskdino_pixel_decoder(feats, sd, p, cfg, be):
    """MaskDINOEncoder.forward_features (maskdino/pixel_decoder/maskdino_encoder.py:368-434), low2high, masks None"""
    f3, f4, f5 = feats["res3"], feats["res4"], feats["res5"]
    extra = gn(conv(f5, sd, p + "input_proj.3.0.", stride=2, padding=1), sd, p + "input_proj.3.1.")
    srcs = [gn(conv(f, sd, "%sinput_proj.%d.0." % (p, i)), sd, "%sinput_proj.%d.1." % (p, i)) for i, f in enumerate((f3, f4, f5))] + [extra]
    B, dev = f3.shape[0], f3.device
    zero = [torch.zeros(B, s.shape[2], s.shape[3], dtype=torch.bool, device=dev) for s in srcs]
    poses = [pos_sine(z, 128, offset=0.0) for z in zero]
    shapes = [tuple(s.shape[-2:]) for s in srcs]
    t = p + "transformer."
    src = torch.cat([s.flatten(2).transpose(1, 2) for s in srcs], 1)
    pos = torch.cat([pe.flatten(2).transpose(1, 2) + sd[t + "level_embed"][i].view(1, 1, -1) for i, pe in enumerate(poses)], 1)
    mask = torch.cat([z.flatten(1) for z in zero], 1)
    refs = encoder_ref_points(shapes, torch.ones(B, 4, 2, device=dev))
    for i in range(cfg["md_enc_layers"]):
        src = encoder_layer(src, pos, refs, shapes, mask, sd, "%sencoder.layers.%d." % (t, i), be)
    outs, st = [], 0
    for (H, W) in shapes:
        outs.append(src[:, st:st + H * W].transpose(1, 2).reshape(B, -1, H, W))
        st += H * W
    cur = gn(F.conv2d(f3, sd[p + "adapter_1.weight"]), sd, p + "adapter_1.norm.")
    y = cur + F.interpolate(outs[0], size=cur.shape[-2:], mode="bilinear", align_corners=False)
    y = F.relu(gn(F.conv2d(y, sd[p + "layer_1.weight"], padding=1), sd, p + "layer_1.norm."))
    mf = F.relu(gn(F.conv_transpose2d(y, sd[p + "mask_features.0.weight"], sd[p + "mask_features.0.bias"], stride=2), sd, p + "mask_features.1."))
    return conv(mf, sd, p + "mask_features.3."), outs


def maskdino_decoder(ms_feats, mask_features, sd, p, cfg, be, dn=None, topk_override=None):
    """MaskDINODecoder.forward in TRAINING mode (maskdino_decoder.py:377-518) + TransformerDecoder.forward (dino_decoder.py:94-168):
    two-stage initialisation from the flattened memory (reversed level order), the proposals' own predictions (interm_outputs), the
    de-noising queries in front (dn = (label queries (B,P,256), boxes (B,P,4) un-sigmoided, self-attention mask) or None), a class / mask /
    box prediction from the initial queries and after every layer.  Returns per-prediction lists (initial + layers) over ALL queries and the
    interm outputs; the caller splits off the de-noising part."""
    nl = len(ms_feats)
    xs = [ms_feats[nl - 1 - i] for i in range(nl)]
    shapes = [tuple(x.shape[-2:]) for x in xs]
    src = torch.cat([x.flatten(2).transpose(1, 2) for x in xs], 1)
    B, dev = src.shape[0], src.device
    mask = torch.zeros(B, src.shape[1], dtype=torch.bool, device=dev)
    vr = torch.ones(B, nl, 2, device=dev)
    om, prop = gen_proposals(src, mask, shapes)
    om = ln(lin(om, sd, p + "enc_output."), sd, p + "enc_output_norm.")
    cls_un = lin(om, sd, p + "class_embed.")
    coord_un = mlp(om, sd, p + "_bbox_embed.", 3) + prop
    nq = cfg["md_num_queries"]
    topk = torch.topk(cls_un.max(-1)[0], nq, dim=1)[1] if topk_override is None else topk_override
    ref_undetach = torch.gather(coord_un, 1, topk.unsqueeze(-1).repeat(1, 1, 4))
    tgt_undetach = torch.gather(om, 1, topk.unsqueeze(-1).repeat(1, 1, om.shape[-1]))

    def heads(x):
        d = ln(x, sd, p + "decoder_norm.")
        return lin(d, sd, p + "class_embed."), be.mask_einsum(mlp(d, sd, p + "mask_embed.", 3), mask_features)
    ic, im = heads(tgt_undetach)
    interm = {"pred_logits": ic, "pred_boxes": ref_undetach.sigmoid(), "pred_masks": im}
    tgt, ref_un, tgt_mask = tgt_undetach.detach(), ref_undetach.detach(), None
    if dn is not None:
        tgt, ref_un, tgt_mask = torch.cat([dn[0], tgt], 1), torch.cat([dn[1], ref_un], 1), dn[2]
    cls0, m0 = heads(tgt)
    classes, masks_ = [cls0], [m0]
    ref = ref_un.sigmoid()
    refs, out, hs = [ref], tgt, []
    for l in range(cfg["md_dec_layers"]):
        ref_in = ref[:, :, None] * torch.cat([vr, vr], -1)[:, None]
        qpos = mlp(sine_embed_4(ref_in[:, :, 0, :]), sd, p + "decoder.ref_point_head.", 2)
        out = decoder_layer(out, qpos, ref_in, src, shapes, mask, sd, "%sdecoder.layers.%d." % (p, l), be, tgt_mask)
        new_ref = (mlp(out, sd, p + "_bbox_embed.", 3) + inverse_sigmoid(ref)).sigmoid()
        ref = new_ref.detach()
        refs.append(new_ref)
        hs.append(ln(out, sd, p + "decoder.norm."))
    for h in hs:
        c, m = heads(h)
        classes.append(c)
        masks_.append(m)
    boxes = [ref_un.sigmoid()] + [(mlp(h, sd, p + "_bbox_embed.", 3) + inverse_sigmoid(r)).sigmoid() for r, h in zip(refs[:-1], hs)]   # pred_box (:357-375)
    return dict(classes=classes, masks=masks_, boxes=boxes, interm=interm, topk=topk)


# ------------------------------------------------------------------------------------------------ CondInst mask branch
def mask_head_small_conv(feats, sd, p):
    """MaskHeadSmallConv.forward, fpns None (ddetrs_dn.py:1633-1689); feats = [s8, s16, s32] NCHW"""
    x = F.relu(conv(feats[-1], sd, p + "lay3.", padding=1))
    x = feats[-2] + F.interpolate(x, size=feats[-2].shape[-2:], mode="nearest")
    x = F.relu(conv(x, sd, p + "lay4.", padding=1))
    x = feats[-3] + F.interpolate(x, size=feats[-3].shape[-2:], mode="nearest")
    x = F.relu(conv(x, sd, p + "jia_dcn.", padding=1))
    return F.relu(conv(F.relu(conv(x, sd, p + "lay1.", padding=1)), sd, p + "lay2.", padding=1))


def backbone_and_projections(x, pad, sd, cfg, be=None):
    """HIPIE_IMG.detr.detr.backbone (MaskedBackbone + Joiner) and input_proj of coco_forward (ddetrs_dn.py:271-320): the three backbone
    levels + the stride-2 extra level whose mask is a resize of the LEVEL-0 mask"""
    p = "detr.detr."
    if cfg.get("backbone", "vit") != "vit":
        raise NotImplementedError("training step: ViT backbones only")
    feats = vit_backbone(x, sd, p + "backbone.0.backbone.", cfg, be)
    names = ["res3", "res4", "res5"]
    fmasks = [down_mask(pad, feats[n].shape[-2:]) for n in names]
    poses = [pos_sine(m, cfg["hidden_dim"] // 2) for m in fmasks]
    srcs = [gn(conv(feats[n], sd, "%sinput_proj.%d.0." % (p, i)), sd, "%sinput_proj.%d.1." % (p, i)) for i, n in enumerate(names)]
    s4 = gn(conv(feats["res5"], sd, p + "input_proj.3.0.", stride=2, padding=1), sd, p + "input_proj.3.1.")
    m4 = down_mask(fmasks[0], s4.shape[-2:])
    return feats, srcs + [s4], fmasks + [m4], poses + [pos_sine(m4, cfg["hidden_dim"] // 2)]
