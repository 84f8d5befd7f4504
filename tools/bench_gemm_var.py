#!/usr/bin/env python3
"""Timing ablations of the split GEMM kernels (library built with `make EXTRA=-DHIPIE_GEMM_VARIANTS`): HIPIE_GEMM_VARIANT =
0 full kernel, 1 no epilogue, 2 no DMA inside the k loop (gemm2 only), 3 no stage barrier / wait (gemm2 only).  Results of variants > 0 are
wrong by construction; only the time is read.  Also: zero operands instead of random ones (the DVFS give-back of MI355X_MICROARCH.md)."""
import os
import sys

import torch

sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

M = 32768


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
    shapes = [("qkv", 1280, 3840), ("fc2", 5120, 1280)]
    for zero in (False, True):
        for name, K, N in shapes:
            x = torch.zeros(M, K, device="cuda") if zero else torch.randn(M, K, device="cuda")
            w = torch.zeros(N, K, device="cuda") if zero else torch.randn(N, K, device="cuda") * K ** -0.5
            xs, ws = ops.to_hl8(x), ops.hl8_pack(w)
            b = torch.randn(N, device="cuda")
            row = []
            for v in os.environ.get("VARIANTS", "0,1,2,3").split(","):
                os.environ["HIPIE_GEMM_VARIANT"] = v
                t = bench(lambda: ops.gemm(xs, ws, b, out_fmt=ops.HL8, split=True))
                row.append("v%s %.3f ms (MFMA %4.0f TF)" % (v, t, 6.0 * M * K * N / t / 1e9))
            print("%s %-4s gemm2=%s: " % ("zeros " if zero else "random", name, os.environ.get("HIPIE_GEMM2", "1")) + "  ".join(row), flush=True)


if __name__ == "__main__":
    main()
