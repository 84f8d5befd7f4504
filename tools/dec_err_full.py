#!/usr/bin/env python3
"""Where does the DINO decoder's error at the headline configuration come from?  Per-layer errors of dec_hs / dec_refs against
tests/golden/stages_full.npz, with optional exact replacements of single operations (diagnosis only; monkeypatches from OUTSIDE the
product):  python tools/dec_err_full.py [mha32] [fixture=e2e_full] [stages=stages_full]"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("HIPIE_MIOPEN_FIND", "0")
import _synth  # noqa: E402
from util import Golden, rel_err  # noqa: E402
from hipie_amd import ops  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402

torch.set_grad_enabled(False)
flags = [a for a in sys.argv[1:] if "=" not in a]
kw = dict(a.split("=") for a in sys.argv[1:] if "=" in a)
fx, stg = kw.get("fixture", "e2e_full"), kw.get("stages", "stages_full")
g, st = Golden(fx), Golden(stg)
model = HIPIE_IMG(HipieConfig.from_dict(g.meta["cfg"]), Precision.split3(), device="cuda")
model.load_state_dict(_synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist")), strict=True)
model.finalize()
model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])

if "mha32" in flags:          # exact fp32 softmax attention instead of the fp16-operand flash kernel (decoder self-attention, BERT, ...)
    real = ops.flash_attn

    def exact(q, k, v, scale, bias_h=None, bias_w=None, key_mask=None, clamp=0.0, out_f32=False):
        if bias_h is not None or bias_w is not None or q.shape[1] > 2048:
            return real(q, k, v, scale, bias_h, bias_w, key_mask, clamp, out_f32)
        B, Nq, H, hd = q.shape
        s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * scale
        if key_mask is not None:
            s = s.masked_fill(~key_mask.bool()[:, None, None, :], float("-inf"))
        o = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), v.float()).reshape(B, Nq, H * hd)
        return o if out_f32 else o.to(q.dtype)
    ops.flash_attn = exact
    import hipie_amd.modeling.transformer as T
    T.ops.flash_attn = exact
if "qkv32" in flags:          # additionally keep q / k / v in fp32 (no fp16 rounding of the attention operands)
    import hipie_amd.modeling.transformer as T
    for m in model.modules():
        if isinstance(m, T.MultiheadAttention):
            m.attn_dtype = torch.float32

if "bi32" in flags:           # exact fp32 bi-directional fusion attention (fuse_helper.py:69-121) instead of the fp16-operand kernels
    import hipie_amd.modeling.transformer as T
    for m in model.modules():
        if isinstance(m, T.BiMultiHeadAttention):
            m.attn_dtype = torch.float32

    def bi_exact(q, k, vv, vl, text_mask, clamp=50000.0, out_f32=False):
        B, Nv, H, hd = q.shape
        L = k.shape[1]
        w = torch.einsum("bnhd,blhd->bhnl", q.float(), k.float()).clamp(-clamp, clamp)
        wT = w.transpose(2, 3)
        wl = (wT - wT.max(-1, keepdim=True)[0]).clamp(-clamp, clamp).softmax(-1)
        wv = w.masked_fill(~text_mask.bool()[:, None, None, :], float("-inf")).softmax(-1)
        ov = torch.einsum("bhnl,blhd->bnhd", wv, vl.float()).reshape(B, Nv, H * hd)
        ol = torch.einsum("bhln,bnhd->blhd", wl, vv.float()).reshape(B, L, H * hd)
        return ov, ol
    T.ops.bi_xattn = bi_exact

if "vit32" in flags:          # exact fp32 ViT attention (materialised scores) on the kernel's operand contract: qkv / tables HL8, q pre-scaled into
    import hipie_amd.modeling.vit as V   # the exp2 domain, tables / scale; output HL8

    def vit_exact(qkv, tab_h, tab_w, grid_hw, heads):
        gh, gw = grid_hw
        B, N, C6 = qkv.shape
        hd = C6 // (6 * heads)
        f = ops.hl8_unpack(qkv).view(B, N, 3, heads, hd)
        q, k, v = (f[:, :, i].permute(0, 2, 1, 3) for i in range(3))                     # (B, heads, N, hd)
        th, tw = ops.hl8_unpack(tab_h), ops.hl8_unpack(tab_w)
        ih = torch.arange(gh, device=qkv.device)[:, None] - torch.arange(gh, device=qkv.device)[None, :] + gh - 1
        iw = torch.arange(gw, device=qkv.device)[:, None] - torch.arange(gw, device=qkv.device)[None, :] + gw - 1
        out = torch.empty(B, N, heads * hd, device=qkv.device)
        for b in range(B):
            for h0 in range(0, heads, 4):
                qq = q[b, h0:h0 + 4]
                s = qq @ k[b, h0:h0 + 4].transpose(-1, -2)
                rq = qq.reshape(-1, gh, gw, hd)
                s = s.view(-1, gh, gw, gh, gw) + torch.einsum("mhwc,hkc->mhwk", rq, th[ih])[..., :, None] \
                    + torch.einsum("mhwc,wkc->mhwk", rq, tw[iw])[..., None, :]
                pr = torch.softmax(s.view(-1, N, N) * 0.6931471805599453, -1)
                out[b, :, h0 * hd:(h0 + 4) * hd] = (pr @ v[b, h0:h0 + 4]).permute(1, 0, 2).reshape(N, -1)
        return ops.hl8_pack(out)
    V.ops.vit_attn_split = vit_exact

imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
ids, mask, pmap = _synth.synth_token_ids(2, g.meta["detection"]["n_classes"], 64, seed=74)
caps = {}
d = model.detr
h1 = d.detr.transformer.decoder.register_forward_hook(lambda m, i, o: caps.__setitem__("dec", o))
h0 = d.detr.transformer.decoder.register_forward_pre_hook(lambda m, i: caps.__setitem__("dec_in", i))
out = model.forward_raw([{"image": im, "task": "detection", "input_ids": ids[i], "attention_mask": mask[i],
                          "positive_map_label_to_token": pmap} for i, im in enumerate(imgs)])
hs, refs = caps["dec"][0], caps["dec"][1]
print("flags", flags, "dec_hs", tuple(hs.shape), "dec_refs", tuple(refs.shape))


def sub_err(key, t, i):
    """error of slice i (first dim) on the fixture's strided subsample of the FULL tensor"""
    full = t.float().cpu().contiguous()
    sub = st.meta.get("subsampled", {})
    want = st[key]
    if key in sub:
        step, shape = sub[key]
        idx = torch.arange(0, full.numel(), step)
        per = full[0].numel()
        sel = (idx // per) == i
        a, b = full.reshape(-1)[idx[sel]], want[sel]
    else:
        a, b = full[i].reshape(-1), want[i].reshape(-1)
    return float((a - b).abs().max() / b.abs().max())


for i in range(hs.shape[0]):
    print("layer %d: hs %.2e" % (i, sub_err("dec_hs", hs, i)))
for i in range(refs.shape[0]):
    print("refs %d: %.2e" % (i, sub_err("dec_refs", refs, i)))
KEYS = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino", "pred_logits_maskdino",
        "pred_boxes_maskdino"]
print(" ".join("%s=%.1e" % (k, rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k])) for k in KEYS))
