#!/usr/bin/env python3
"""diagnostics of hipie_vit_attn_split on the golden attention cases: error vs the fp64 formulation and vs variants of it."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from util import Golden, rel_err
import oracle.ops as oo
from hipie_amd import ops
from hipie_amd.modeling.vit import resize_rel_pos
import test_gpu_kernels as tk
torch.set_grad_enabled(False)
g = Golden("vit_attn")
for name in ["window14", "global16", "global64", "global_rect"]:
    c, sd, x = tk.vit_attn_case(g, name)
    B, H, W, C = x.shape; heads = c["heads"]; hd = C // heads; scale = hd ** -0.5
    c1 = scale * ops.LOG2E
    qkv = tk._vit_qkv(c, sd, x)
    f = qkv.clone(); f[..., :C] *= c1
    th, tw = resize_rel_pos(H, sd["rel_pos_h"]), resize_rel_pos(W, sd["rel_pos_w"])
    q, k, v = qkv.double().reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4).reshape(3, B * heads, H * W, hd).unbind(0)
    want = oo.vit_attention_core(q, k, v, th.double(), tw.double(), (H, W), scale)
    want = want.view(B, heads, H * W, hd).permute(0, 2, 1, 3).reshape(B, H * W, C).float()
    def run(ff, tth, ttw):
        o = ops.vit_attn_split(ff.cuda(), tth.cuda(), ttw.cuda(), (H, W), heads)
        return ops.hl8_unpack(o).cpu()
    P = ops.hl8_pack
    def zero_lo(t):      # HL8 with the lo halves zeroed
        z = P(t).reshape(*t.shape[:-1], t.shape[-1] // 8, 2, 8).clone()
        z[..., 1, :] = 0
        return z.reshape(*t.shape[:-1], 2 * t.shape[-1])
    full = run(P(f), P(th / scale), P(tw / scale))
    nolo = run(zero_lo(f), zero_lo(th / scale), zero_lo(tw / scale))
    # lo zeroed only in v
    fz = P(f).reshape(B, H * W, 3, C // 8, 2, 8).clone(); fz[:, :, 2, :, 1, :] = 0
    vnolo = run(fz.reshape(B, H * W, 6 * C), P(th / scale), P(tw / scale))
    e = rel_err(full, want)
    idx = (full - want).abs().argmax()
    tok, ch = int(idx) // C % (H * W), int(idx) % C
    print("%-12s err %.2e (worst token %d ch %d: got %.6f want %.6f) | all lo zeroed %.2e | v lo zeroed %.2e | mean signed err %.2e, rms %.2e" %
          (name, e, tok, ch, full.reshape(-1)[idx], want.reshape(-1)[idx], rel_err(nolo, want), rel_err(vnolo, want),
           float((full - want).mean() / want.abs().max()), float((full - want).pow(2).mean().sqrt() / want.abs().max())))
    # per-channel-group error profile (d index within head)
    d = (full - want).abs().reshape(B, H * W, heads, hd).amax((0, 1, 2)) / want.abs().max()
    print("   per-d max err: " + " ".join("%.0e" % float(t) for t in d[::8]))
    tq = (full - want).abs().reshape(B, H * W, C).amax((0, 2)) / want.abs().max()
    print("   per-token max err (first 8 / last 8): " + " ".join("%.0e" % float(t) for t in tq[:8]) + " ... " + " ".join("%.0e" % float(t) for t in tq[-8:]))
