"""Timings of the training-side operator backwards at the training geometry of a 1024^2 image, two images per GPU:
  python tools/bench_train_ops.py
MSDA backward has its own line in tools/bench_msda.py bwd."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hipie_amd import ops  # noqa: E402


def timeit(fn, n=10):
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
    B, Q, C, H, W = 2, 300, 256, 256, 256                       # MaskDINO: 300 queries, mask features at stride 4
    e = torch.randn(B, Q, C, generator=g).to(dev)
    f = torch.randn(B, C, H, W, generator=g).to(dev)
    go = torch.randn(B, Q, H, W, generator=g).to(dev)
    fw = timeit(lambda: ops.mask_einsum(e, f, precision=1))
    bw = timeit(lambda: ops.mask_einsum_backward(e, f, go))
    fl = 2.0 * B * Q * C * H * W
    print("mask_einsum B=%d Q=%d C=%d %dx%d: forward %.3f ms, backward %.3f ms (%.1f / %.1f TFLOP/s algorithmic)"
          % (B, Q, C, H, W, fw, bw, fl / fw / 1e9, 2 * fl / bw / 1e9))
    B, Q, H, W = 2, 300, 128, 128                                # CondInst head: matched instances, mask features at stride 8, x2 output
    feats = torch.randn(B, 8, H, W, generator=g).to(dev)
    refs = (torch.rand(B * Q, 2, generator=g) * 1024).to(dev)
    params = (torch.randn(B * Q, 169, generator=g) * 0.1).to(dev)
    go = torch.randn(B * Q, 2 * H, 2 * W, generator=g).to(dev)
    fw = timeit(lambda: ops.dynamic_mask(feats, refs, params, Q, stride=8, up=2))
    bw = timeit(lambda: ops.dynamic_mask_backward(feats, refs, params, go, Q, stride=8, up=2))
    print("dynamic_mask B=%d Q=%d %dx%d up=2: forward (fp32 kernel) %.3f ms, backward %.3f ms; grad_out read %.1f MB -> %.2f TB/s"
          % (B, Q, H, W, fw, bw, go.numel() * 4 / 1e6, go.numel() * 4 / bw / 1e9))


if __name__ == "__main__":
    main()
