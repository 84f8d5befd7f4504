#!/usr/bin/env python3
"""what do the wrong MSDA groups look like? (see tools/msda_stress.py)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd import ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)


def rn(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


B, S, Q = 8, 21760, 300
shapes = torch.tensor([[128, 128], [64, 64], [32, 32], [16, 16]], device=dev)
lstart = torch.tensor([0, 16384, 20480, 21504], device=dev)
val = rn(B, S, 8, 32)
dref4 = torch.rand(B, Q, 4, 4, generator=g).to(dev) * 0.5 + 0.25
doff = rn(B, Q, 8, 4, 4, 2)
dlog = rn(B, Q, 8, 16)
M = 174080
bres = rn(M, 256)
bw = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
fn = lambda: ops.msda_fused(val, shapes, lstart, dref4, doff, dlog)
ref = fn().clone().view(B, Q, 8, 32)
torch.cuda.synchronize()
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
seen = 0
for it in range(40):
    side.wait_stream(main)
    for _ in range(3):
        ops.gemm(bres, bw, None, split=True, out_fmt=ops.F32)
    with torch.cuda.stream(side):
        outs = [fn() for _ in range(4)]
    main.wait_stream(side)
    torch.cuda.synchronize()
    for o in outs:
        o = o.view(B, Q, 8, 32)
        d = (o - ref).abs().amax(-1)
        for b, q, h in torch.nonzero(d > 0).tolist():
            got, want = o[b, q, h], ref[b, q, h]
            ratio = got / want
            # is it the correct result of ANOTHER group?
            match = torch.nonzero(((ref - got).abs().amax(-1)) == 0)
            nbad_lanes = int(((got - want).abs().view(8, 4).amax(-1) > 0).sum())
            print("(b %d, q %3d, head %d): %d of 8 lanes wrong | ratio min %.4f max %.4f | finite %s | equals ref group %s | got[:4] %s want[:4] %s" % (
                b, q, h, nbad_lanes, float(ratio.min()), float(ratio.max()), bool(torch.isfinite(got).all()), match.tolist()[:2],
                [round(v, 4) for v in got[:4].tolist()], [round(v, 4) for v in want[:4].tolist()]), flush=True)
            seen += 1
        if seen > 24:
            break
    if seen > 24:
        break
print("groups examined:", seen)
