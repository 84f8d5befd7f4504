#!/usr/bin/env python3
"""hipie_vit_attn_split at the bench geometry (B=8, 64x64 tokens, 16 heads x 80) and the windowed form (200 windows of 14x14)."""
import sys
import torch
sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402
B, H, W, heads, hd = 8, 64, 64, 16, 80
C = heads * hd


def bench(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


qkv = ops.to_hl8(torch.randn(B, H * W, 3 * C, device="cuda") * 0.8)
th, tw = ops.hl8_pack(torch.randn(2 * H - 1, hd) * 0.2).cuda(), ops.hl8_pack(torch.randn(2 * W - 1, hd) * 0.2).cuda()
t = bench(lambda: ops.vit_attn_split(qkv, th, tw, (H, W), heads))
gf = 4.0 * (H * W) ** 2 * C * B / 1e9
print("global 64x64 B=8: %.3f ms  %.0f TFLOP/s algorithmic, %.0f MFMA-issued (QK x3, PV x3)" % (t, gf / t, 3.0 * gf / t))
if len(sys.argv) > 1 and sys.argv[1] == "global":        # PMC passes: the global kernel only (the kernel-name filter matches both forms)
    sys.exit(0)
q2 = ops.to_hl8(torch.randn(200, 196, 3 * C, device="cuda") * 0.8)
t2h, t2w = ops.hl8_pack(torch.randn(27, hd) * 0.2).cuda(), ops.hl8_pack(torch.randn(27, hd) * 0.2).cuda()
t2 = bench(lambda: ops.vit_attn_split(q2, t2h, t2w, (14, 14), heads))
print("windows 14x14 x200: %.3f ms  %.0f TFLOP/s algorithmic" % (t2, 4.0 * 196 * 196 * C * 200 / 1e9 / t2))
# the 96-slot instance: 84 x 84 token grid (1344-pixel images, BASELINE configs[4]), one image
H3 = W3 = 84
q3 = ops.to_hl8(torch.randn(1, H3 * W3, 3 * C, device="cuda") * 0.8)
t3h, t3w = ops.hl8_pack(torch.randn(2 * H3 - 1, hd) * 0.2).cuda(), ops.hl8_pack(torch.randn(2 * W3 - 1, hd) * 0.2).cuda()
t3 = bench(lambda: ops.vit_attn_split(q3, t3h, t3w, (H3, W3), heads))
print("global 84x84 B=1: %.3f ms  %.0f TFLOP/s algorithmic" % (t3, 4.0 * (H3 * W3) ** 2 * C / 1e9 / t3))
