import sys, os, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
os.environ.setdefault("HIPIE_MIOPEN_FIND", "0")
import torch
torch.set_grad_enabled(False)
import _synth
from util import Golden, rel_err
from hipie_amd.config import HipieConfig, Precision
from hipie_amd.hipie_img import HIPIE_IMG
KEYS = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino", "pred_logits_maskdino", "pred_boxes_maskdino"]
g = Golden("e2e_tiny")
def run(prec, tag):
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    model = HIPIE_IMG(cfg, prec, device="cuda")
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()})
    model.load_state_dict(sd, strict=True); model.finalize()
    imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
    ids, mask, pmap = _synth.synth_token_ids(2, g.meta["detection"]["n_classes"], 64, seed=74)
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    out = model.forward_raw([{"image": im, "task": "detection", "input_ids": ids[i], "attention_mask": mask[i], "positive_map_label_to_token": pmap} for i, im in enumerate(imgs)])
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("%-28s max %.1e | " % (tag, max(errs.values())) + " ".join("%s=%.1e" % (k.replace("pred_", ""), v) for k, v in errs.items()), flush=True)
f16, f32 = torch.float16, torch.float32
base = Precision.fast()
run(base, "fast")
run(dataclasses.replace(base, head=f32), "fast head=f32")
run(dataclasses.replace(base, act=f32), "fast act=f32")
run(dataclasses.replace(base, value=f32), "fast value=f32")
run(dataclasses.replace(base, gemm=f32), "fast gemm(vit)=f32")
run(dataclasses.replace(base, head=f32, act=f32), "fast head=f32 act=f32")
run(dataclasses.replace(base, head=f32, act=f32, value=f32), "fast head,act,value=f32")
run(dataclasses.replace(base, act=f32, einsum=0), "fast act=f32 einsum=0")
run(dataclasses.replace(base, text=f32), "fast text=f32")
run(dataclasses.replace(base, attn_fast=False), "fast attn exact")
run(dataclasses.replace(base, gemm=f32, text=f32), "fast gemm(vit),text=f32")
