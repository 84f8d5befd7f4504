#!/usr/bin/env python3
"""Does a hand-written kernel give the same bits when ANOTHER kernel runs beside it on a second stream?  Each op of the decoder / head path
is run alone (reference bits), then repeatedly on a side stream while heavy kernels of the encoder path (split GEMM, fused FFN, MSDA,
LayerNorm) run on the main stream.  Any difference is a concurrency hazard of that op (shared scratch, unordered LDS / global access)."""
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


# ---- background (main stream) work: the encoder-side kernels ------------------------------------------------------------------------
M = 174080
bx = ops.to_hl8(rn(M, 256))
bw = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
bb = rn(256)
lin1 = torch.nn.Linear(256, 2048).to(dev)
lin2 = torch.nn.Linear(2048, 256).to(dev)
lnw, lnb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
bres = rn(M, 256)
B, S = 8, 21760
shapes = torch.tensor([[128, 128], [64, 64], [32, 32], [16, 16]], device=dev)
lstart = torch.tensor([0, 16384, 20480, 21504], device=dev)
bval = rn(B, S, 8, 32)
bref = torch.rand(B, S, 4, 2, generator=g).to(dev)
boff = rn(B, S, 8, 4, 4, 2)
blog = rn(B, S, 8, 16)
vx = ops.to_hl8(rn(32768, 1280))
vw = ops.hl8_pack(rn(5120, 1280, scale=0.03)).to(dev)


def background():
    ops.gemm(bx, bw, bb, split=True, out_fmt=ops.F32)
    ops.ffn_fused(bx.view(B, S, 512), lin1, lin2)
    ops.msda_fused(bval, shapes, lstart, bref, boff, blog)
    ops.add_layernorm_dec(bres, bres, lnw, lnb, 1e-5, "hl8", want16=True)
    ops.gemm(vx, vw, None, split=True, out_fmt=ops.HL8, act=ops.ACT_GELU)


# ---- ops under test (side stream): the decoder / head kernels -------------------------------------------------------------------------
Q = 300
t32 = rn(B, Q, 256)
w_s = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
w_f = ops.hl8_pack(rn(2048, 256, scale=0.06)).to(dev)
w_b = ops.hl8_pack(rn(256, 2048, scale=0.02)).to(dev)
hid = rn(B, Q, 2048)
qkv = rn(B, Q, 3, 8, 32)
dref = torch.rand(B, Q, 4, 4, generator=g).to(dev) * 0.5 + 0.25
doff = rn(B, Q, 8, 4, 4, 2)
dlog = rn(B, Q, 8, 16)
emb = rn(B, Q, 256)
feats = rn(B, 256, 256, 256)
head = torch.nn.Sequential()
mlp2 = type("M", (), {})()
big_src = rn(B, S, 256)
w_v = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)

TESTS = {
    "gemm_small f32->f32 (K 256)": lambda: ops.gemm(t32, w_s, bb, split=True, out_fmt=ops.F32),
    "gemm_small f32->hl8 relu (N 2048)": lambda: ops.gemm(t32, w_f, None, split=True, out_fmt=ops.HL8, act=ops.ACT_RELU),
    "gemm_small f32->f32 (K 2048)": lambda: ops.gemm(hid, w_b, bb, split=True, out_fmt=ops.F32),
    "gemm value_proj f32 rows (M 174080)": lambda: ops.gemm(big_src, w_v, bb, split=True, out_fmt=ops.F32),
    "attn_split hd 32": lambda: ops.attn_split(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], 32 ** -0.5),
    "msda_fused decoder (4-d refs)": lambda: ops.msda_fused(bval, shapes, lstart, dref, doff, dlog),
    "add_layernorm_dec": lambda: ops.add_layernorm_dec(t32, emb, lnw, lnb, 1e-5, "hl8", want16=True)[0],
    "sine_embed": lambda: ops.sine_embed(dref[:, :, 0, :]),
    "box_refine": lambda: ops.box_refine(rn_box, dref[:, :, 0, :].contiguous()),
    "mask_einsum (ws)": lambda: ops.mask_einsum(emb, feats, precision=1),
    "to_hl8": lambda: ops.to_hl8(t32),
    "torch cat + gather + add": lambda: torch.gather(torch.cat([t32, emb], 1) + 1.0, 1, gidx),
    "torch matmul (fold)": lambda: emb @ fold_w,
}
rn_box = rn(B, Q, 4)
gidx = torch.randint(0, 2 * Q, (B, 100, 256), generator=g).to(dev)
fold_w = rn(256, 256, scale=0.06)

side = torch.cuda.Stream()
main = torch.cuda.current_stream()
bad_total = 0
for name, fn in TESTS.items():
    ref = fn()
    ref = (ref[0] if isinstance(ref, tuple) else ref).clone()
    torch.cuda.synchronize()
    solo = fn()
    solo = solo[0] if isinstance(solo, tuple) else solo
    torch.cuda.synchronize()
    solo_ok = torch.equal(solo, ref)
    bad = 0
    worst = 0.0
    for it in range(12):
        side.wait_stream(main)
        background()
        with torch.cuda.stream(side):
            outs = []
            for _ in range(6):
                o = fn()
                outs.append(o[0] if isinstance(o, tuple) else o)
        main.wait_stream(side)
        torch.cuda.synchronize()
        for o in outs:
            if not torch.equal(o, ref):
                bad += 1
                worst = max(worst, float((o.float() - ref.float()).abs().max()))
    bad_total += bad
    print("%-40s solo repeat %s | beside other kernels: %d / 72 differ%s" % (name, "identical" if solo_ok else "DIFFERS", bad, "" if not bad else "  (max |diff| %.3e)" % worst), flush=True)
print("CONCURRENCY STRESS: %s" % ("clean" if bad_total == 0 else "%d differing results" % bad_total))
