#!/usr/bin/env python3
"""GPU vs CPU gradients of intermediate tensors of the training step (same product code on both devices)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

import test_training as TT  # noqa: E402
from hipie_amd.training import net  # noqa: E402

res = {}
for dev in ("cpu", "cuda"):
    z, meta, model, step, batch, targets = TT._train_step_case(dev)
    stash = {}
    orig_mb = step.mask_branch

    def mb(memory, shapes, sd, _o=orig_mb, _s=stash):
        memory.retain_grad()
        _s["memory"] = memory
        lv = _o(memory, shapes, sd)
        lv.retain_grad()
        _s["mask_feats"] = lv
        return lv
    step.mask_branch = mb
    orig_msc = net.mask_head_small_conv

    def msc(feats, sd, p, _s=stash):
        outs = []
        import torch.nn.functional as F
        x = F.relu(net.conv(feats[-1], sd, p + "lay3.", padding=1)); x.retain_grad(); _s["a_lay3"] = x
        x = feats[-2] + F.interpolate(x, size=feats[-2].shape[-2:], mode="nearest")
        x = F.relu(net.conv(x, sd, p + "lay4.", padding=1)); x.retain_grad(); _s["b_lay4"] = x
        x = feats[-3] + F.interpolate(x, size=feats[-3].shape[-2:], mode="nearest")
        x = F.relu(net.conv(x, sd, p + "jia_dcn.", padding=1)); x.retain_grad(); _s["c_dcn"] = x
        x = F.relu(net.conv(x, sd, p + "lay1.", padding=1)); x.retain_grad(); _s["d_lay1"] = x
        return F.relu(net.conv(x, sd, p + "lay2.", padding=1))
    net.mask_head_small_conv = msc
    orig_bp = net.backbone_and_projections

    def bp(x, pad, sd, cfg, _s=stash):
        r = orig_bp(x, pad, sd, cfg)
        _s["_padmask"] = torch.cat([m.flatten(1) for m in r[2]], 1).detach().cpu()
        return r
    net.backbone_and_projections = bp
    with torch.enable_grad():
        losses = step.loss_dict(batch, targets)
        total = sum(losses.values())
        total.backward()
    net.mask_head_small_conv = orig_msc
    net.backbone_and_projections = orig_bp
    padmask = stash.pop("_padmask")
    res[dev] = {k: (v.detach().cpu(), v.grad.detach().cpu()) for k, v in stash.items()}
    res[dev]["P"] = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None and "mask_head" in n}
for k in sorted(res["cpu"]):
    if k == "P":
        continue
    a, b = res["cpu"][k], res["cuda"][k]
    print("%-12s value diff %.2e | grad diff %.2e (rel to max |grad| %.3e)" % (k, float((a[0] - b[0]).abs().max() / a[0].abs().max()),
          float((a[1] - b[1]).abs().max() / a[1].abs().max()), float(a[1].abs().max())))
a, b = res["cpu"]["memory"][0], res["cuda"]["memory"][0]
d = (a - b).abs().amax(-1) / a.abs().max()
print("memory value diff: valid tokens %.2e (%d), padded tokens %.2e (%d)" % (float(d[~padmask].max()), int((~padmask).sum()), float(d[padmask].max()), int(padmask.sum())))
ga, gb = res["cpu"]["memory"][1], res["cuda"]["memory"][1]
gd = (ga - gb).abs().amax(-1) / ga.abs().max()
print("memory grad diff: valid %.2e padded %.2e" % (float(gd[~padmask].max()), float(gd[padmask].max())))
for n in res["cpu"]["P"]:
    a, b = res["cpu"]["P"][n], res["cuda"]["P"][n]
    print("%-40s param grad diff %.2e" % (n, float((a - b).abs().max() / a.abs().max())))
