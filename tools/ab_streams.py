#!/usr/bin/env python3
"""A/B of the stream overlap (DDETRSegmUniDN.use_streams: the text encoder on a side stream beside the backbone) on the timed workload:
same process, same model, same batch; alternating blocks of steps with the side stream on and off, and a comparison of the a22 outputs of
both modes against the run-to-run spread of each (the step is not bit-reproducible from run to run: ~1e-6 relative, both modes alike)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402
from hipie_amd.postprocess import inference_compact  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
cfg = HipieConfig.vit_huge()
torch.manual_seed(0)
model = HIPIE_IMG(cfg, Precision.split3(), device=dev)
bench.randomize_degenerate_inits(model)
model.finalize()
batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)


def step():
    return inference_compact(model, model.forward_raw(batch), batch, topk=100)


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def snap():
    o = model.forward_raw(batch)
    torch.cuda.synchronize()
    return {k: v.clone() for k, v in o.items() if torch.is_tensor(v)}


def cmp(a, b, what):
    bad = [(k, float((a[k].float() - b[k].float()).abs().max())) for k in a if not torch.equal(a[k], b[k])]
    print("%-44s %s" % (what, "bit-identical" if not bad else "DIFFER: " + ", ".join("%s %.2e" % kv for kv in bad)), flush=True)


model.detr.use_streams = False
for _ in range(2):
    step()
fg, md = model.last_topk()
model.pin_topk(fg.clone(), md.clone())            # the two-stage selections pinned: a 1-ulp difference cannot re-order the queries
off1, off2 = snap(), snap()
cmp(off1, off2, "streams off, run 1 vs run 2")
model.detr.use_streams = True
step()
step()
ons = [snap() for _ in range(4)]
for i in range(1, 4):
    cmp(ons[0], ons[i], "streams on, run 1 vs run %d" % (i + 1))
cmp(ons[0], off1, "streams on vs off")
model.pin_topk(None, None)
for rnd in range(3):
    for on in (True, False):
        model.detr.use_streams = on
        step()
        print("round %d  streams %-5s  %.2f ms / step" % (rnd, "on" if on else "off", timed(5)), flush=True)
