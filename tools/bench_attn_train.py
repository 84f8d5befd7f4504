#!/usr/bin/env python3
"""The fused training attention (csrc/attn_train.hip, training/functions.FusedAttentionFunction) against the materialised torch formulation at the
training step's global-block shape: 2 images x 16 heads, 64 x 64 tokens, q' / k' of 208 columns (80 + 64 + 64), v of 80.  Forward and
forward + backward, ms per ViT block; the GPU kernels of the fused path alone (operand conversions excluded) are listed by rocprofv3 if run under it."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hipie_amd.training.functions import FusedAttentionFunction  # noqa: E402

BH, N, DQ, DV = int(os.environ.get("BH", 32)), 4096, 208, 80
g = torch.Generator(device="cuda").manual_seed(0)
qa = (torch.randn(BH, N, DQ, device="cuda", generator=g) * 0.5).requires_grad_(True)
ka = torch.randn(BH, N, DQ, device="cuda", generator=g).requires_grad_(True)
v = torch.randn(BH, N, DV, device="cuda", generator=g).requires_grad_(True)
go = torch.randn(BH, N, DV, device="cuda", generator=g) * 1e-3


def materialised():
    return (qa @ ka.transpose(-2, -1)).softmax(dim=-1) @ v


def fused():
    return FusedAttentionFunction.apply(qa, ka, v)


def bench(fn, backward, n=5):
    for _ in range(2):
        o = fn()
        if backward:
            o.backward(go)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        o = fn()
        if backward:
            o.backward(go)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for name, fn in (("materialised (library fp32 GEMMs + softmax)", materialised), ("fused (hipie_attn_train_*)", fused)):
    f = bench(fn, False)
    fb = bench(fn, True)
    torch.cuda.reset_peak_memory_stats()
    fn().backward(go)
    torch.cuda.synchronize()
    print("%-46s forward %.2f ms, forward + backward %.2f ms per block (BH = %d, N = %d); peak memory of one block %.2f GB"
          % (name, f, fb, BH, N, torch.cuda.max_memory_allocated() / 2 ** 30))
