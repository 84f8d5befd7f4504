#!/usr/bin/env python3
"""GPU time of the main stages of one bench forward and of every hand-written kernel class.  Stage rows: the device is SYNCHRONISED in
front of and behind each stage's forward and the host clock is read there, so a row is that stage's own GPU time --
the rows of nested stages add up (round 4 bracketed the stages with HIP events on a stream the host was ~1000 launches ahead of; the
`backbone` row of profiles/r04_stage_times.txt came out as the whole forward).  The synchronisations cost the overlap between stages,
so the sum sits slightly above the free-running forward, which is printed first.  Output: gpurun_out/stage_times.txt"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402


def main():
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    cfg = HipieConfig.vit_huge()
    pol = sys.argv[1] if len(sys.argv) > 1 else "split3"
    model = HIPIE_IMG(cfg, getattr(Precision, pol)(), device=dev)
    bench.randomize_degenerate_inits(model)
    model.finalize()
    batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)
    d = model.detr
    stages = {"text_encoder (BERT)": model.text_encoder[0], "backbone (ViT-H + SFP)": d.detr.backbone[0],
              "input_proj": d.detr.input_proj, "DINO encoder (6 layers + VL fusion)": d.detr.transformer.encoder,
              "DINO decoder (6 layers)": d.detr.transformer.decoder, "DINO transformer total": d.detr.transformer,
              "MaskDINO pixel decoder": d.mask_dino.pixel_decoder, "MaskDINO decoder + einsums": d.mask_dino.predictor,
              "mask_head (CondInst convs)": d.mask_head, "whole DDETRSegmUniDN": d}
    rec = {k: [] for k in stages}

    import time
    hooks_on = [False]

    def pre(name):
        def f(m, a, kw=None):
            if hooks_on[0]:
                torch.cuda.synchronize()
                rec[name].append([time.perf_counter(), None])
        return f

    def post(name):
        def f(m, a, out):
            if hooks_on[0]:
                torch.cuda.synchronize()
                rec[name][-1][1] = time.perf_counter()
        return f
    for k, m in stages.items():
        if isinstance(m, torch.nn.ModuleList):          # input_proj: time each member, summed
            for sub in m:
                sub.register_forward_pre_hook(pre(k))
                sub.register_forward_hook(post(k))
        else:
            m.register_forward_pre_hook(pre(k))
            m.register_forward_hook(post(k))
    for _ in range(3):
        model.forward_raw(batch)
    torch.cuda.synchronize()
    n = 3
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        model.forward_raw(batch)                      # free-running (hooks inert)
    t1.record()
    torch.cuda.synchronize()
    hooks_on[0] = True
    for _ in range(n):
        model.forward_raw(batch)
    hooks_on[0] = False
    os.makedirs("gpurun_out", exist_ok=True)
    from hipie_amd import ops
    ops.PROFILE.shapes = len(sys.argv) > 2 and sys.argv[2] == "shapes"
    ops.PROFILE.enable("all")
    model.forward_raw(batch)
    prof = ops.PROFILE.summary()
    ops.PROFILE.disable()
    with open("gpurun_out/stage_times.txt", "w") as f:
        f.write("policy %s: forward_raw %.2f ms free-running (mean of %d)\n" % (pol, t0.elapsed_time(t1) / n, n))
        for tag, (mean, cnt, tot) in sorted(prof.items(), key=lambda kv: -kv[1][2]):
            f.write("   kernel class %-18s n=%4d mean=%8.3f ms total=%8.2f ms\n" % (tag, cnt, mean, tot))
        for k, v in rec.items():
            f.write("%-40s %8.2f ms  (%d calls / forward; synchronised boundaries)\n" % (k, sum((b - a) * 1e3 for a, b in v) / n, len(v) // n))
    print(open("gpurun_out/stage_times.txt").read())


if __name__ == "__main__":
    main()
