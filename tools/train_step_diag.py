#!/usr/bin/env python3
"""which operator moves the mask-head gradients of the training step on the GPU?  The step with the HIP backend, then with ONE operator at a
time swapped for a plain-torch formulation on the same device; gradient errors against tests/golden/train_step_tiny.npz."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import test_training as TT  # noqa: E402
from hipie_amd.training import net  # noqa: E402


def torch_dynamic_mask(mask_feats, ref_points, params, num_insts, stride, up):
    B, C, H, W = mask_feats.shape
    dev = mask_feats.device
    xs = torch.arange(0, W * stride, stride, dtype=torch.float32, device=dev) + stride // 2
    ys = torch.arange(0, H * stride, stride, dtype=torch.float32, device=dev) + stride // 2
    n_all = ref_points.shape[0]
    p = params
    w0, w1, w2 = p[:, :(C + 2) * 8].reshape(n_all, 8, C + 2), p[:, 80:144].reshape(n_all, 8, 8), p[:, 144:152].reshape(n_all, 1, 8)
    b0, b1, b2 = p[:, 152:160], p[:, 160:168], p[:, 168:169]
    outs, st = [], 0
    for b, n in enumerate(num_insts):
        r = ref_points[st:st + n]
        relx = r[:, 0].view(n, 1, 1) - xs.view(1, 1, W).expand(n, H, W)
        rely = r[:, 1].view(n, 1, 1) - ys.view(1, H, 1).expand(n, H, W)
        x0 = torch.cat([relx[:, None], rely[:, None], mask_feats[b][None].expand(n, C, H, W)], 1).reshape(n, C + 2, H * W)
        x1 = F.relu(torch.bmm(w0[st:st + n], x0) + b0[st:st + n, :, None])
        x2 = F.relu(torch.bmm(w1[st:st + n], x1) + b1[st:st + n, :, None])
        outs.append((torch.bmm(w2[st:st + n], x2) + b2[st:st + n, :, None]).reshape(n, 1, H, W))
        st += n
    t = torch.cat(outs, 0)
    if up > 1:
        h, w = t.shape[2:]
        t = F.pad(t, pad=(0, 1, 0, 1), mode="replicate")
        t = F.interpolate(t, size=(up * h + 1, up * w + 1), mode="bilinear", align_corners=True)
        t = F.pad(t, pad=(up // 2, 0, up // 2, 0), mode="replicate")[:, :, :up * h, :up * w]
    return t[:, 0]


def run(variant):
    z, meta, model, step, batch, targets = TT._train_step_case("cuda")

    class BE(net.HipBackend):
        pass
    if variant == "torch dynamic_mask":
        BE.dynamic_mask = staticmethod(torch_dynamic_mask)
    if variant == "torch mask_einsum":
        BE.mask_einsum = staticmethod(lambda e, f: torch.einsum("bqc,bchw->bqhw", e, f))
    if variant == "no MIOpen find":
        torch.backends.cudnn.benchmark = False
    if variant == "convs as unfold + matmul":
        orig = net.conv

        def conv_unfold(x, sd, p, stride=1, padding=0):
            w = sd[p + "weight"]
            if w.shape[-1] == 3 and stride == 1:
                cols = F.unfold(x, 3, padding=padding)
                y = torch.matmul(w.reshape(w.shape[0], -1), cols).view(x.shape[0], w.shape[0], x.shape[2], x.shape[3])
                b = sd.get(p + "bias")
                return y if b is None else y + b.view(1, -1, 1, 1)
            return orig(x, sd, p, stride, padding)
        net.conv = conv_unfold
    step.be = BE
    with torch.enable_grad():
        losses = step.loss_dict(batch, targets)
        total = sum(losses.values())
        total.backward()
    steps = json.loads(bytes(z["grad_steps"]).decode())
    params = dict(model.named_parameters(remove_duplicate=False))
    errs = []
    for k in z.files:
        if k.startswith("grad/"):
            name = k[5:]
            p = params[name]
            g = (torch.zeros_like(p) if p.grad is None else p.grad).reshape(-1).cpu()
            if name in steps:
                g = g[::steps[name]]
            w = torch.from_numpy(z[k])
            errs.append((float((g - w).abs().max() / (w.abs().max() + 1e-12)), name))
    errs.sort(reverse=True)
    print("%-28s total %.5f | worst gradients: %s" % (variant, float(total), ", ".join("%s %.1e" % (n.replace("detr.", ""), e) for e, n in errs[:5])), flush=True)
    if variant == "convs as unfold + matmul":
        net.conv = orig


for v in ("hip backend", "torch dynamic_mask", "torch mask_einsum", "no MIOpen find", "convs as unfold + matmul"):
    run(v)
