#!/usr/bin/env python3
"""micro-benchmark of the HBM-bound head kernels at the bench geometry (B = 8, 1024^2): the mask-logit contraction
(300 x 256 x 256^2) and the dynamic mask head (910 instances), with their algorithmic bytes per launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hipie_amd import ops  # noqa: E402


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
    g = torch.Generator().manual_seed(0)
    B, Q, C, H = 8, 300, 256, 256
    emb = torch.randn(B, Q, C, generator=g).to(dev)
    feat = torch.randn(B, C, H, H, generator=g).to(dev)
    for prec, od, wsp in ((1, torch.float32, True), (1, torch.float32, False), (1, torch.bfloat16, True), (2, torch.bfloat16, True)):
        t = bench(lambda: ops.mask_einsum(emb, feat, precision=prec, out_dtype=od, workspace=wsp))
        by = feat.numel() * 4 + emb.numel() * 4 + B * Q * H * H * (4 if od == torch.float32 else 2)
        print("mask_einsum precision %d out %s %s: %.3f ms  %.2f TB/s algorithmic  %.0f TFLOP/s" % (
            prec, str(od).split(".")[-1], "workspace (LDS-DMA)" if wsp else "no workspace", t, by / t / 1e9, 2.0 * B * Q * C * H * H / t / 1e9))
    for dt in (torch.float16, torch.bfloat16):
        f16 = feat.to(dt)
        for split in (True, False):
            t = bench(lambda: ops.mask_einsum16(emb, f16, split=split))
            by = f16.numel() * 2 + emb.numel() * 2 * (2 if split else 1) + B * Q * H * H * 2
            print("mask_einsum16 %s split=%d: %.3f ms  %.2f TB/s algorithmic  %.0f TFLOP/s" % (
                str(dt).split(".")[-1], split, t, by / t / 1e9, 2.0 * B * Q * C * H * H / t / 1e9))
    nq = 910
    feats = torch.randn(B, 8, 128, 128, generator=g).to(dev)
    refs = (torch.rand(B * nq, 2, generator=g) * 1024).to(dev)
    params = torch.randn(B * nq, 169, generator=g).to(dev)
    for od in (torch.float32, torch.bfloat16):
        t = bench(lambda: ops.dynamic_mask(feats, refs, params, nq, stride=8, up=2, out_dtype=od), n=10)
        by = B * nq * 256 * 256 * (4 if od == torch.float32 else 2)
        print("dynamic_mask out %s: %.3f ms  %.2f TB/s written" % (str(od).split(".")[-1], t, by / t / 1e9))
    for od in (torch.float32, torch.float16):
        t = bench(lambda: ops.dynamic_mask(feats, refs, params, nq, stride=8, up=2, out_dtype=od, mlp_dtype=torch.float16), n=10)
        by = B * nq * 256 * 256 * (4 if od == torch.float32 else 2)
        print("dynamic_mask16 (f16 MFMA layers) out %s: %.3f ms  %.2f TB/s written" % (str(od).split(".")[-1], t, by / t / 1e9))


if __name__ == "__main__":
    main()
