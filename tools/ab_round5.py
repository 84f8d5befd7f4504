#!/usr/bin/env python3
"""Same-box A/B of the round-5 kernel changes: run once with the in-tree library and once with HIPIE_LIB_PATH=<another build> and compare.
Prints ms per launch of the global / windowed split attention and of the four ViT-H linears in the forms the timed step uses them
(qkv -> HL8, proj -> fp32 + residual, fc1 -> GELU -> HL8, fc2 -> fp32 + residual), B = 8 at 1024 x 1024."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd import _lib, ops  # noqa: E402


def bench(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    print("library:", _lib.LIB_PATH)
    torch.manual_seed(0)
    B, H, W, heads, hd = 8, 64, 64, 16, 80
    C = heads * hd
    qkv = ops.to_hl8(torch.randn(B, H * W, 3 * C, device="cuda") * 0.8)
    th, tw = ops.hl8_pack(torch.randn(2 * H - 1, hd) * 0.2).cuda(), ops.hl8_pack(torch.randn(2 * W - 1, hd) * 0.2).cuda()
    for rep in range(2):
        t = bench(lambda: ops.vit_attn_split(qkv, th, tw, (H, W), heads))
        gf = 4.0 * (H * W) ** 2 * C * B / 1e9
        print("attn global 64x64 B=8: %.4f ms  %.0f TFLOP/s algorithmic" % (t, gf / t), flush=True)
    q2 = ops.to_hl8(torch.randn(200, 196, 3 * C, device="cuda") * 0.8)
    t2h, t2w = ops.hl8_pack(torch.randn(27, hd) * 0.2).cuda(), ops.hl8_pack(torch.randn(27, hd) * 0.2).cuda()
    t2 = bench(lambda: ops.vit_attn_split(q2, t2h, t2w, (14, 14), heads))
    print("attn windows 14x14 x200: %.4f ms" % t2, flush=True)
    del qkv, q2
    M = 32768
    x = ops.to_hl8(torch.randn(M, 1280, device="cuda"))
    x4 = ops.to_hl8(torch.randn(M, 5120, device="cuda"))
    res = torch.randn(M, 1280, device="cuda")
    for name, a, K, N, kw in (("qkv  -> hl8", x, 1280, 3840, dict(out_fmt=ops.HL8)),
                              ("proj -> f32 + resid", x, 1280, 1280, dict(out_fmt=ops.F32, resid=res)),
                              ("fc1  -> gelu -> hl8", x, 1280, 5120, dict(out_fmt=ops.HL8, act=ops.ACT_GELU)),
                              ("fc1  -> hl8 (no act)", x, 1280, 5120, dict(out_fmt=ops.HL8)),
                              ("fc2  -> f32 + resid", x4, 5120, 1280, dict(out_fmt=ops.F32, resid=res))):
        w = ops.hl8_pack(torch.randn(N, K, device="cuda") * K ** -0.5)
        b = torch.randn(N, device="cuda")
        t = bench(lambda: ops.gemm(a, w, b, split=True, **kw), n=20)
        print("gemm %-22s M=%d K=%d N=%d: %.4f ms  %.0f TFLOP/s algorithmic" % (name, M, K, N, t, 2.0 * M * K * N / t / 1e9), flush=True)
    # GELU accuracy of the fc1 epilogue against fp64 on a small problem
    xs = torch.randn(512, 256, device="cuda") * 2
    ws = torch.randn(320, 256, device="cuda") * 0.2
    bs = torch.randn(320, device="cuda")
    y = ops.gemm(ops.to_hl8(xs), ops.hl8_pack(ws), bs, split=True, out_fmt=ops.F32, act=ops.ACT_GELU)
    ref = torch.nn.functional.gelu(xs.double() @ ws.double().t() + bs.double())
    print("gelu epilogue: max abs err %.3e (|ref| max %.2f)" % (float((y.double() - ref).abs().max()), float(ref.abs().max())))


if __name__ == "__main__":
    main()
