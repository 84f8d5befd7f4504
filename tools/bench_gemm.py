#!/usr/bin/env python3
"""library-GEMM check for the ViT-H linears (M = batch*4096 tokens): default hipBLASLt heuristic vs PyTorch TunableOp's
pick, per shape.  Writes the tuned table to gpurun_out/tunableop_vit.csv."""
import os
import sys
import time

import torch
import torch.nn.functional as F

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
SHAPES = [("qkv", 1280, 3840), ("proj", 1280, 1280), ("fc1", 1280, 5120), ("fc2", 5120, 1280)]


def bench(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    dev = "cuda"
    res = {}
    for tuned in (False, True):
        if tuned:
            os.makedirs("gpurun_out", exist_ok=True)
            torch.cuda.tunable.set_filename("gpurun_out/tunableop_vit.csv")
            torch.cuda.tunable.set_max_tuning_duration(30)
            torch.cuda.tunable.set_max_tuning_iterations(20)
            torch.cuda.tunable.enable(True)
            torch.cuda.tunable.tuning_enable(True)
        for name, K, N in SHAPES:
            x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
            w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
            b = torch.randn(N, device=dev, dtype=torch.bfloat16)
            ms = bench(lambda: F.linear(x, w, b))
            res[(name, tuned)] = ms
            print("%-5s tuned=%d  %.3f ms  %.0f TFLOP/s" % (name, tuned, ms, 2.0 * M * K * N / ms / 1e9), flush=True)
    tot0 = sum(res[(n, False)] for n, _, _ in SHAPES)
    tot1 = sum(res[(n, True)] for n, _, _ in SHAPES)
    print("per block: default %.3f ms, tuned %.3f ms  (x32 blocks: %.1f -> %.1f ms)" % (tot0, tot1, 32 * tot0, 32 * tot1))


if __name__ == "__main__":
    main()
