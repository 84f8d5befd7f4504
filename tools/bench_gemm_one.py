#!/usr/bin/env python3
"""one shape of hipie_gemm, a few launches (for PMC passes): bench_gemm_one.py [qkv|fc2] [split|plain]"""
import sys
import torch
sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402
shape = sys.argv[1] if len(sys.argv) > 1 else "qkv"
split = (sys.argv[2] if len(sys.argv) > 2 else "split") == "split"
M = 32768
K, N = {"qkv": (1280, 3840), "fc2": (5120, 1280), "fc1": (1280, 5120), "proj": (1280, 1280)}[shape]
x = torch.randn(M, K, device="cuda")
w = torch.randn(N, K, device="cuda") * K ** -0.5
b = torch.randn(N, device="cuda")
xs, ws = (ops.to_hl8(x), ops.hl8_pack(w)) if split else (x.half(), w.half())
for _ in range(6):
    ops.gemm(xs, ws, b, out_fmt=ops.F32, split=split)
torch.cuda.synchronize()
