#!/usr/bin/env python3
"""Same-box, same-process A/B of host-level formulation switches on the timed step (ViT-H, bs 8, 1024^2, split3): alternates the step with a
switch off / on (A B A B ...) and prints the mean ms of each.  Switches (functions the model code consults per call):
  ln     ops.split_linear_ln_ok          output_proj + residual + norm1 of the encoder layers as ONE launch (hipie_gemm_ln)
  dv     transformer.decoder_split_values  the decoder value projections as one batched thin-K GEMM
  bi     ops.bi_i2t_folded_ok            the vision-language fusion attention without the visual projections of width embed_dim
  gn     ops.group_norm(out_nchw=...)    the mask_features GroupNorm writes NCHW itself (no transposing copy behind it)
python tools/ab_step.py [ln,dv] [rounds]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from hipie_amd import ops  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402
from hipie_amd.postprocess import inference_compact  # noqa: E402
import hipie_amd.modeling.transformer as T  # noqa: E402
import hipie_amd.modeling.maskdino as MD  # noqa: E402


def main():
    which = (sys.argv[1] if len(sys.argv) > 1 else "ln,dv").split(",")
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    torch.set_grad_enabled(False)
    dev = torch.device("cuda", 0)
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, Precision.split3(), device=dev)
    bench.randomize_degenerate_inits(model)
    model.finalize()
    batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev, seed=0)
    keep = {"bi": ops.bi_i2t_folded_ok, "ln": ops.split_linear_ln_ok, "dv": T.decoder_split_values, "gn": ops.group_norm}

    def gn_plain(*a, **kw):
        kw.pop("out_nchw", None)
        return keep["gn"](*a, **kw)

    def switch(on):
        if "ln" in which:
            ops.split_linear_ln_ok = keep["ln"] if on else (lambda *a: False)
        if "bi" in which:
            ops.bi_i2t_folded_ok = keep["bi"] if on else (lambda *a: False)
        if "gn" in which:
            ops.group_norm = keep["gn"] if on else gn_plain
        if "dv" in which:
            T.decoder_split_values = MD.decoder_split_values = keep["dv"] if on else (lambda *a: False)

    def step():
        return inference_compact(model, model.forward_raw(batch), batch, topk=100)

    def timed(n=5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(n):
            out = step()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / n, out

    res = {False: [], True: []}
    outs = {}
    for on in (False, True):
        switch(on)
        for _ in range(3):
            step()
    for r in range(rounds):
        for on in (False, True):
            switch(on)
            t, outs[on] = timed()
            res[on].append(t)
    switch(True)
    d = (outs[True].float() - outs[False].float()).abs().max().item()
    print("switches %s: off %s -> mean %.2f ms;  on %s -> mean %.2f ms;  compact predictions max |diff| %.2e" % (
        which, ["%.2f" % t for t in res[False]], sum(res[False]) / rounds, ["%.2f" % t for t in res[True]], sum(res[True]) / rounds, d))


if __name__ == "__main__":
    main()
