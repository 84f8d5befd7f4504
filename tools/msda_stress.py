#!/usr/bin/env python3
"""narrow-down of the MSDA concurrency hazard found by tools/concurrency_stress.py"""
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
eref = torch.rand(B, S, 4, 2, generator=g).to(dev)
eoff = rn(B, S, 8, 4, 4, 2)
elog = rn(B, S, 8, 16)
dref4 = torch.rand(B, Q, 4, 4, generator=g).to(dev) * 0.5 + 0.25
dref2 = torch.rand(B, Q, 4, 2, generator=g).to(dev)
doff = rn(B, Q, 8, 4, 4, 2)
dlog = rn(B, Q, 8, 16)
M = 174080
bx = ops.to_hl8(rn(M, 256))
bw = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
vx = ops.to_hl8(rn(32768, 1280))
vw = ops.hl8_pack(rn(1280, 1280, scale=0.03)).to(dev)
lnw, lnb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
t32 = rn(8, 300, 256)
bw384 = ops.hl8_pack(rn(384, 256, scale=0.06)).to(dev)
lin1 = torch.nn.Linear(256, 2048).to(dev)
lin2 = torch.nn.Linear(2048, 256).to(dev)
emb = rn(8, 300, 256)
feats = rn(8, 256, 256, 256)
bx2048 = ops.to_hl8(rn(21760, 2048))
bw2048 = ops.hl8_pack(rn(256, 2048, scale=0.02)).to(dev)
bwN2048 = ops.hl8_pack(rn(2048, 256, scale=0.06)).to(dev)
bres = rn(M, 256)

BG = {
    "none": lambda: None,
    "gemm K256": lambda: ops.gemm(bx, bw, None, split=True, out_fmt=ops.F32),
    "gemm K256 f32 rows": lambda: ops.gemm(bres, bw, None, split=True, out_fmt=ops.F32),
    "gemm_small (M 2400)": lambda: [ops.gemm(t32, bw, None, split=True, out_fmt=ops.F32) for _ in range(40)],
    "gemm k256 thin (N 384)": lambda: ops.gemm(bx, bw384, None, split=True, out_fmt=ops.F32),
    "ffn_fused": lambda: ops.ffn_fused(bx.view(8, 21760, 512), lin1, lin2),
    "layernorm dec": lambda: ops.add_layernorm_dec(bres, bres, lnw, lnb, 1e-5, "hl8", want16=True),
    "to_hl8": lambda: ops.to_hl8(bres),
    "mask_einsum": lambda: ops.mask_einsum(emb, feats, precision=1),
    "gemm<256> K 2048": lambda: ops.gemm(bx2048, bw2048, None, split=True, out_fmt=ops.F32),
    "gemm<256> N 2048 K 256": lambda: ops.gemm(bx, bwN2048, None, split=True, out_fmt=ops.HL8),
}
uloc = torch.rand(B, Q, 8, 4, 4, 2, generator=g).to(dev)
uatt = torch.rand(B, Q, 8, 4, 4, generator=g).to(dev) / 16
FG = {

    "decoder form, 4-d refs": lambda: ops.msda_fused(val, shapes, lstart, dref4, doff, dlog),
    "encoder form (Lq = S)": lambda: ops.msda_fused(val, shapes, lstart, eref, eoff, elog),
}
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
for fname, fn in FG.items():
    ref = fn().clone()
    torch.cuda.synchronize()
    for bname, bg in BG.items():
        bad, worst, where = 0, 0.0, None
        for it in range(10):
            side.wait_stream(main)
            for _ in range(3):
                bg()
            with torch.cuda.stream(side):
                outs = [fn() for _ in range(4)]
            main.wait_stream(side)
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    d = (o - ref).abs()
                    worst = max(worst, float(d.max()))
                    if where is None:
                        idx = torch.nonzero(d.view(B, -1, 8, 32).amax(-1) > 0)
                        where = "%d (b,q,head) groups differ, first %s, last %s, heads %s" % (
                            idx.shape[0], idx[0].tolist(), idx[-1].tolist(), sorted(set(idx[:, 2].tolist())))
        print("%-26s beside %-14s: %2d / 40 differ%s" % (fname, bname, bad, "" if not bad else "  max %.2e; %s" % (worst, where)), flush=True)
