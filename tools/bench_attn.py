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


def run_rel(B=8, g=64, heads=16, hd=80, dt=torch.bfloat16, iters=30, fast=True, win=False):
    """hipie_vit_attn_rel: global (g x g grid) or, win=True, the 14x14 windows of the same batch (25 windows per image)."""
    gen = torch.Generator().manual_seed(0)
    if win:
        B, g = B * 25, 14
    N, C = g * g, heads * hd
    qkv = torch.randn(B, N, 3 * C, generator=gen)
    qkv[..., :C] *= hd ** -0.5 * ops.LOG2E
    qkv = qkv.to(dt).cuda()
    th = (torch.randn(2 * g - 1, hd, generator=gen) * 0.3 * hd ** 0.5).to(dt).cuda()
    tw = (torch.randn(2 * g - 1, hd, generator=gen) * 0.3 * hd ** 0.5).to(dt).cuda()
    ops.vit_attn_rel(qkv, th, tw, (g, g), heads, fast=fast)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        ops.vit_attn_rel(qkv, th, tw, (g, g), heads, fast=fast)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    return ms, 4.0 * N * N * C * B / ms / 1e9


if __name__ == "__main__":
    if os.environ.get("REL") == "4":          # per-phase cycle counters of the software-pipelined kernel (timing build)
        import ctypes
        from hipie_amd import _lib
        os.environ["HIPIE_VA_ABL"] = os.environ.get("TIMING_ABL", "8")
        ms, tf = run_rel()
        buf = (ctypes.c_ulonglong * 16)()
        lib = _lib.load()
        lib.hipie_va_debug.argtypes = [ctypes.c_void_p]
        print("rc", lib.hipie_va_debug(buf), "ms", ms)
        for g in range(2):
            v = list(buf[8 * g:8 * g + 6])
            nt = max(v[5], 1)
            print("wave %d: per tile cycles: phase1 %.0f  phase2 %.0f  barrier %.0f  grow-rate %.3f  total/tile %.0f (nt %d)" % (
                4 * g, v[0] / nt, v[1] / nt, v[2] / nt, v[3] / nt, v[4] / nt, nt))
        sys.exit(0)
    if os.environ.get("REL") == "3":          # timing ablations (library built with -DHIPIE_VA_ABLATIONS)
        for rnd in range(2):
            for abl in [int(x) for x in os.environ.get('ABLS', '0,1,2,3,4,5,6').split(',')]:
                os.environ["HIPIE_VA_ABL"] = str(abl)
                ms, tf = run_rel()
                print("round %d  ABL=%d  %.4f ms" % (rnd, abl, ms), flush=True)
        sys.exit(0)
    if os.environ.get("REL") == "2":          # one variant only (PMC passes)
        ms, tf = run_rel(dt=torch.float16 if os.environ.get("DT") == "f16" else torch.bfloat16, fast=os.environ.get("EXACT") != "1")
        print("rel  %.4f ms  %.1f TFLOP/s" % (ms, tf), flush=True)
        sys.exit(0)
    if os.environ.get("REL") == "1":
        for rnd in range(3):          # interleaved rounds of the variants in one process
            ms, tf = run_fused()
            print("round %d  fused(old)      %.4f ms  %.1f TFLOP/s" % (rnd, ms, tf), flush=True)
            ms, tf = run_rel()
            print("round %d  rel bf16 fast   %.4f ms  %.1f TFLOP/s" % (rnd, ms, tf), flush=True)
            ms, tf = run_rel(dt=torch.float16)
            print("round %d  rel f16 fast    %.4f ms  %.1f TFLOP/s" % (rnd, ms, tf), flush=True)
            ms, tf = run_rel(dt=torch.float16, fast=False)
            print("round %d  rel f16 exact   %.4f ms  %.1f TFLOP/s" % (rnd, ms, tf), flush=True)
        ms, tf = run_rel(win=True)
        print("windows rel bf16 fast   %.4f ms  %.1f TFLOP/s" % (ms, tf), flush=True)
        ms, tf = run_rel(g=84, B=2)
        print("84x84 B=2 rel bf16 fast %.4f ms  %.1f TFLOP/s" % (ms, tf), flush=True)
        sys.exit(0)
    if os.environ.get("FUSED_ONLY") == "1":
        ms, tf = run_fused()
        print("fused (prio=%s)  %.4f ms  %.1f TFLOP/s" % (os.environ.get("HIPIE_FA_PRIO", "0"), ms, tf), flush=True)
        sys.exit(0)
    for waves in ("8", "4"):
        os.environ["HIPIE_FA_WAVES"] = waves
        ms, tf, err = run()
        print("WAVES=%s  %.3f ms  %.1f TFLOP/s  relerr %.2e" % (waves, ms, tf, err), flush=True)
