"""Kernel-level restatements (test infrastructure -- see oracle/__init__.py).

Every function here is the checker for one hand-written HIP kernel of hipie_amd/csrc/.
All arithmetic is fp32 (or fp64 when asked) PyTorch on the CPU.
"""
import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- MSDeformAttn sampling (a10)
def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights):
    """Multi-scale deformable attention sampling.

    Restates the CUDA kernel ``ms_deformable_im2col_gpu_kernel`` + ``ms_deform_attn_im2col_bilinear``
    (hipie/models/deformable_detr/ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 and :33-84) with explicit
    corner gathers instead of F.grid_sample: pixel coordinate = loc*size - 0.5; a sample contributes only when
    -1 < h,w < size; each of the 4 corners contributes only when it lies inside the map.

    value (B,S,M,D), spatial_shapes (L,2) [(H,W)], sampling_locations (B,Lq,M,L,P,2) as (x,y) in [0,1],
    attention_weights (B,Lq,M,L,P)  ->  (B,Lq,M*D)
    """
    B, S, M, D = value.shape
    _, Lq, _, L, P, _ = sampling_locations.shape
    out = value.new_zeros(B, Lq, M, D)
    start = 0
    bidx = torch.arange(B).view(B, 1, 1, 1)
    midx = torch.arange(M).view(1, 1, M, 1)
    for lvl in range(L):
        H, W = int(spatial_shapes[lvl][0]), int(spatial_shapes[lvl][1])
        v = value[:, start:start + H * W]                       # (B, H*W, M, D)
        loc = sampling_locations[:, :, :, lvl]                   # (B,Lq,M,P,2)
        w_im = loc[..., 0] * W - 0.5
        h_im = loc[..., 1] * H - 0.5
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h_low = torch.floor(h_im)
        w_low = torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        acc = value.new_zeros(B, Lq, M, P, D)
        for dy, dx, wgt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            yy, xx = h_low + dy, w_low + dx
            ok = (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1) & inside
            idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))          # (B,Lq,M,P)
            g = v[bidx, idx, midx]                                        # (B,Lq,M,P,D)
            acc = acc + g * (wgt * ok.to(value.dtype)).unsqueeze(-1)
        out = out + (acc * attention_weights[:, :, :, lvl].unsqueeze(-1)).sum(3)
        start += H * W
    return out.reshape(B, Lq, M * D)


# --------------------------------------------------------------------------- ViT attention (a4, a5)
def get_rel_pos(q_size, k_size, rel_pos):
    """hipie/backbone/utils.py:63-93 (table linearly re-interpolated when its length != 2*max-1)."""
    max_rel_dist = int(2 * max(q_size, k_size) - 1)
    if rel_pos.shape[0] != max_rel_dist:
        r = F.interpolate(rel_pos.reshape(1, rel_pos.shape[0], -1).permute(0, 2, 1), size=max_rel_dist, mode="linear")
        r = r.reshape(-1, max_rel_dist).permute(1, 0)
    else:
        r = rel_pos
    q_coords = torch.arange(q_size)[:, None] * max(k_size / q_size, 1.0)
    k_coords = torch.arange(k_size)[None, :] * max(q_size / k_size, 1.0)
    rel = (q_coords - k_coords) + (k_size - 1) * max(q_size / k_size, 1.0)
    return r[rel.long()]


def vit_attention_core(q, k, v, rel_pos_h, rel_pos_w, hw, scale):
    """softmax(scale*q.k + q.Rh[hq,hk] + q.Rw[wq,wk]) v  for one token grid of size hw=(H,W).

    hipie/backbone/vit.py:72-80 and hipie/backbone/utils.py:96-125: the bias is computed from the UNSCALED q.
    q,k,v: (BH, H*W, hd) -> (BH, H*W, hd)
    """
    H, W = hw
    attn = (q * scale) @ k.transpose(-2, -1)
    Rh = get_rel_pos(H, H, rel_pos_h)
    Rw = get_rel_pos(W, W, rel_pos_w)
    BH, _, hd = q.shape
    r_q = q.reshape(BH, H, W, hd)
    rel_h = torch.einsum("bhwc,hkc->bhwk", r_q, Rh)
    rel_w = torch.einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(BH, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(BH, H * W, H * W)
    attn = attn.softmax(dim=-1)
    return attn @ v


def vit_attention(x, sd, prefix, num_heads):
    """Attention.forward, hipie/backbone/vit.py:67-83.  x (B,H,W,C) -> (B,H,W,C)."""
    B, H, W, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[prefix + "qkv.weight"], sd[prefix + "qkv.bias"])
    qkv = qkv.reshape(B, H * W, 3, num_heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.reshape(3, B * num_heads, H * W, -1).unbind(0)
    o = vit_attention_core(q, k, v, sd[prefix + "rel_pos_h"], sd[prefix + "rel_pos_w"], (H, W), hd ** -0.5)
    o = o.view(B, num_heads, H, W, -1).permute(0, 2, 3, 1, 4).reshape(B, H, W, -1)
    return F.linear(o, sd[prefix + "proj.weight"], sd[prefix + "proj.bias"])


# --------------------------------------------------------------------------- VL fusion (a9)
def bi_attention_core(q, k, vv, vl, text_mask):
    """Bi-directional cross attention of BiMultiHeadAttention.forward, hipie/models/deformable_detr/
    fuse_helper.py:69-121, for already projected / head-split tensors.

    q (BH,Nv,hd) (already scaled), k (BH,L,hd), vv (BH,Nv,hd), vl (BH,L,hd), text_mask (B,L) int {0,1}.
    returns out_v (BH,Nv,hd), out_l (BH,L,hd).
    """
    BH, Nv, hd = q.shape
    B = text_mask.shape[0]
    heads = BH // B
    L = k.shape[1]
    w = torch.bmm(q, k.transpose(1, 2))
    w = torch.clamp(w, min=-50000)
    w = torch.clamp(w, max=50000)
    wT = w.transpose(1, 2)
    wl = wT - torch.max(wT, dim=-1, keepdim=True)[0]
    wl = torch.clamp(wl, min=-50000)
    wl = torch.clamp(wl, max=50000)
    wl = wl.softmax(dim=-1)
    # int64 mask: 0 -> -9e15, 1 stays 1 (a uniform +1 shift of the valid logits), fuse_helper.py:97-108
    am = text_mask.to(torch.int64)[:, None, None, :].expand(B, 1, Nv, L)
    am = am.masked_fill(am == 0, int(-9e15))
    w = (w.view(B, heads, Nv, L) + am).view(BH, Nv, L)
    wv = F.softmax(w, dim=-1)
    return torch.bmm(wv, vl), torch.bmm(wl, vv)


def bi_attention_block(v, l, text_mask, sd, prefix, num_heads=8):
    """BiAttentionBlockForCheckpoint.forward (fuse_helper.py:170-179) + BiMultiHeadAttention.forward (:54-139)."""
    p = prefix
    v = F.layer_norm(v, v.shape[-1:], sd[p + "layer_norm_v.weight"], sd[p + "layer_norm_v.bias"], 1e-5)
    l = F.layer_norm(l, l.shape[-1:], sd[p + "layer_norm_l.weight"], sd[p + "layer_norm_l.bias"], 1e-5)
    a = p + "attn."
    B, Nv, _ = v.shape
    L = l.shape[1]
    E = sd[a + "v_proj.weight"].shape[0]
    hd = E // num_heads

    def split(t, n):
        return t.view(B, n, num_heads, hd).transpose(1, 2).reshape(B * num_heads, n, hd)
    q = split(F.linear(v, sd[a + "v_proj.weight"], sd[a + "v_proj.bias"]) * hd ** -0.5, Nv)
    k = split(F.linear(l, sd[a + "l_proj.weight"], sd[a + "l_proj.bias"]), L)
    vv = split(F.linear(v, sd[a + "values_v_proj.weight"], sd[a + "values_v_proj.bias"]), Nv)
    vl = split(F.linear(l, sd[a + "values_l_proj.weight"], sd[a + "values_l_proj.bias"]), L)
    ov, ol = bi_attention_core(q, k, vv, vl, text_mask)
    ov = ov.view(B, num_heads, Nv, hd).transpose(1, 2).reshape(B, Nv, E)
    ol = ol.view(B, num_heads, L, hd).transpose(1, 2).reshape(B, L, E)
    dv = F.linear(ov, sd[a + "out_v_proj.weight"], sd[a + "out_v_proj.bias"])
    dl = F.linear(ol, sd[a + "out_l_proj.weight"], sd[a + "out_l_proj.bias"])
    return v + sd[p + "gamma_v"] * dv, l + sd[p + "gamma_l"] * dl


# --------------------------------------------------------------------------- mask-logit contraction (a21)
def mask_einsum(mask_embed, mask_features):
    """torch.einsum("bqc,bchw->bqhw"), hipie/models/maskdino/transformer_decoder/maskdino_decoder.py:527."""
    return torch.einsum("bqc,bchw->bqhw", mask_embed, mask_features)


# --------------------------------------------------------------------------- CondInst dynamic mask head (a19)
def aligned_bilinear(t, factor):
    """hipie/models/ddetrs_dn.py:1832-1854."""
    if factor == 1:
        return t
    h, w = t.shape[2:]
    t = F.pad(t, pad=(0, 1, 0, 1), mode="replicate")
    oh, ow = factor * h + 1, factor * w + 1
    t = F.interpolate(t, size=(oh, ow), mode="bilinear", align_corners=True)
    t = F.pad(t, pad=(factor // 2, 0, factor // 2, 0), mode="replicate")
    return t[:, :, :oh - 1, :ow - 1]


def dynamic_mask(mask_feats, ref_points, params, num_insts, stride=8, up=2):
    """dynamic_mask_with_coords + parse_dynamic_params + mask_heads_forward + compute_locations,
    hipie/models/ddetrs_dn.py:1411-1502, 1806-1829, 1390-1408, 1857-1870, written per instance instead of as
    one grouped conv:  per instance i of image b,
        x0 = [ref_x - (stride*col + stride//2), ref_y - (stride*row + stride//2), feats_b (8ch)]   (10 ch)
        x1 = relu(W0 x0 + b0) (8), x2 = relu(W1 x1 + b1) (8), y = W2 x2 + b2 (1), then aligned_bilinear(y, up).

    mask_feats (B,8,H,W); ref_points (1, sum(num_insts), 2) pixels (x,y); params (1, sum, 169)
    -> (1, sum, up*H, up*W)
    """
    B, C, H, W = mask_feats.shape
    n_all = ref_points.shape[1]
    xs = torch.arange(0, W * stride, stride, dtype=torch.float32) + stride // 2
    ys = torch.arange(0, H * stride, stride, dtype=torch.float32) + stride // 2
    p = params.reshape(n_all, -1)
    w0, w1, w2 = p[:, :(C + 2) * 8].reshape(n_all, 8, C + 2), p[:, 80:144].reshape(n_all, 8, 8), p[:, 144:152].reshape(n_all, 1, 8)
    b0, b1, b2 = p[:, 152:160], p[:, 160:168], p[:, 168:169]
    outs, st = [], 0
    for b, n in enumerate(num_insts):
        r = ref_points[0, st:st + n]                                        # (n,2)
        relx = r[:, 0].view(n, 1, 1) - xs.view(1, 1, W).expand(n, H, W)
        rely = r[:, 1].view(n, 1, 1) - ys.view(1, H, 1).expand(n, H, W)
        x0 = torch.cat([relx[:, None], rely[:, None], mask_feats[b][None].expand(n, C, H, W)], 1).reshape(n, C + 2, H * W)
        x1 = F.relu(torch.bmm(w0[st:st + n], x0) + b0[st:st + n, :, None])
        x2 = F.relu(torch.bmm(w1[st:st + n], x1) + b1[st:st + n, :, None])
        y = torch.bmm(w2[st:st + n], x2) + b2[st:st + n, :, None]
        outs.append(y.reshape(n, 1, H, W))
        st += n
    logits = torch.cat(outs, 0)
    logits = aligned_bilinear(logits, up)
    return logits.reshape(1, n_all, logits.shape[-2], logits.shape[-1])
