#!/usr/bin/env python3
"""Micro-benchmark of hipie_vit_attn at the ViT-H global-attention shape (B=8, 64x64 tokens, 16 heads x 80):
prints ms / launch and TFLOP/s for the variants selected with HIPIE_FA_WAVES, and checks the result on a
B=1 slice against a plain fp32 torch computation."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hipie_amd import ops


def run(B=8, g=64, heads=16, hd=80, dt=torch.bfloat16, iters=20):
    N, C = g * g, heads * hd
    gen = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, N, 3 * C, generator=gen)).to(dt).cuda()
    rel_h = torch.randn(B * heads, g, N, generator=gen).cuda()          # key-row major
    rel_w = torch.randn(B * heads, N, g, generator=gen).cuda()
    out = ops.vit_attn(qkv, rel_h, rel_w, (g, g), heads, hd ** -0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.vit_attn(qkv, rel_h, rel_w, (g, g), heads, hd ** -0.5)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    fl = 4.0 * N * N * C * B
    # correctness on batch 0, head 0..1
    q, k, v = qkv[0].float().view(N, 3, heads, hd).permute(1, 2, 0, 3)[:, :2]
    s = (q * hd ** -0.5) @ k.transpose(-1, -2)
    s = s.view(2, N, g, g) + rel_h[:2].transpose(1, 2).reshape(2, N, g, 1) + rel_w[:2].view(2, N, 1, g)
    ref = (s.view(2, N, N).softmax(-1) @ v).permute(1, 0, 2).reshape(N, 2 * hd)
    err = float((out[0, :, :2 * hd].float() - ref).abs().max() / ref.abs().max())
    return ms, fl / ms / 1e9, err


def run_fused(B=8, g=64, heads=16, hd=80, dt=torch.bfloat16, iters=30):
    N, C = g * g, heads * hd
    gen = torch.Generator().manual_seed(0)
    qkv = (torch.randn(B, N, 3 * C, generator=gen)).to(dt).cuda()
    th = (torch.randn(2 * g - 1, hd, generator=gen) * 0.3).to(dt).cuda()
    tw = (torch.randn(2 * g - 1, hd, generator=gen) * 0.3).to(dt).cuda()
    ops.vit_attn_fused(qkv, th, tw, (g, g), heads, hd ** -0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.vit_attn_fused(qkv, th, tw, (g, g), heads, hd ** -0.5)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    return ms, 4.0 * N * N * C * B / ms / 1e9


if __name__ == "__main__":
    if os.environ.get("FUSED_ONLY") == "1":
        ms, tf = run_fused()
        print("fused (prio=%s)  %.4f ms  %.1f TFLOP/s" % (os.environ.get("HIPIE_FA_PRIO", "0"), ms, tf), flush=True)
        sys.exit(0)
    for waves in ("8", "4"):
        os.environ["HIPIE_FA_WAVES"] = waves
        ms, tf, err = run()
        print("WAVES=%s  %.3f ms  %.1f TFLOP/s  relerr %.2e" % (waves, ms, tf, err), flush=True)
