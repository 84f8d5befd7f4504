#!/usr/bin/env python3
"""where the MaskCLIP time of a bs-8 post-processing call goes (synchronised wall time per phase): the image-token pass, the patch-visibility
maps of the two fusion sites, the mask-token rows, the rest of HIPIE_IMG.inference.  ViT-L/14-336, random weights, bench.py's batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hipie_amd import open_vocab, postprocess  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402
from hipie_amd.modeling.transformer import set_split  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg = HipieConfig.vit_huge()
torch.manual_seed(0)
model = HIPIE_IMG(cfg, Precision.split3(), device=dev)
bench.randomize_degenerate_inits(model)
model.finalize()
batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)
model.clip = open_vocab.MaskCLIP("ViT-L-14-336", tokenize=bench.synthetic_clip_tokenizer()).to(dev).eval()
model.clip.loaded = True
set_split(model.clip, True)
model.train_labels = [{"id": i, "name": "train%d,alias%d" % (i, i)} for i in range(133)]
names = [{"id": i, "name": ("train%d" % i) if i % 2 else ("novel%d" % i)} for i in range(len(batch[0]["positive_map_label_to_token"]))]
cbatch = [dict(b, open_seg_labels=names) for b in batch]
model.enable_clip = True
out = model.forward_raw(cbatch)
acc = {}


def wrap(obj, name, label=None):
    f = getattr(obj, name)

    def g(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize()
        acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, g)


if os.environ.get("NOWRAP") == "1":                     # free-running calls only (for a kernel trace: tools/top_dispatches.py <trace> 0.09s)
    for _ in range(3):
        postprocess.inference(model, out, cbatch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    postprocess.inference(model, out, cbatch)
    torch.cuda.synchronize()
    print("inference with MaskCLIP, bs 8, free-running: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
    sys.exit(0)
wrap(model.clip, "encode_images")
wrap(model.clip, "blocked_patches")
wrap(model.clip, "mask_rows")
wrap(postprocess, "_sem_pan")
wrap(torch.nn.functional, "interpolate", "F.interpolate (all: the x4 up-sampling of the sem/pan masks and the 336^2 resizes)")
postprocess.inference(model, out, cbatch)
acc.clear()
torch.cuda.synchronize()
t0 = time.perf_counter()
postprocess.inference(model, out, cbatch)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("inference with MaskCLIP, bs 8: %.1f ms (with the phase synchronisations)" % (tot * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-90s %7.1f ms" % (k, v * 1e3))
