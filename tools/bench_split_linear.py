#!/usr/bin/env python3
"""SplitLinearFunction (hipie_gemm forward + backward) against F.linear (hipBLASLt fp32) at the training step's ViT-H shapes"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hipie_amd.training.functions import SplitLinearFunction  # noqa: E402


def t(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for M in (8192, 32768):
    for name, K, N in (("qkv", 1280, 3840), ("proj", 1280, 1280), ("fc1", 1280, 5120), ("fc2", 5120, 1280), ("enc ffn1", 256, 2048)):
        x = torch.randn(M, K, device="cuda", requires_grad=True)
        w = torch.randn(N, K, device="cuda", requires_grad=True)
        b = torch.randn(N, device="cuda", requires_grad=True)
        go = torch.randn(M, N, device="cuda")
        owner = type("_W", (), {})()
        f_split = lambda: SplitLinearFunction.apply(x, w, b, owner, "w")          # noqa: E731
        f_lib = lambda: torch.nn.functional.linear(x, w, b)                      # noqa: E731
        with torch.enable_grad():
            ys, yl = f_split(), f_lib()
            bs = lambda: torch.autograd.grad(ys, (x, w, b), go, retain_graph=True)          # noqa: E731
            bl = lambda: torch.autograd.grad(yl, (x, w, b), go, retain_graph=True)          # noqa: E731
            print("M %5d %-8s K %4d N %4d: forward split %.3f ms / library %.3f ms | backward split %.3f ms / library %.3f ms" % (
                M, name, K, N, t(f_split), t(f_lib), t(bs), t(bl)), flush=True)
