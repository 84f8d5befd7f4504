#!/usr/bin/env python3
"""micro-benchmark of the bi-directional VL-fusion attention (hipie_bi_xattn: two flash passes, 8 heads x 256) at the bench
geometry (B = 8, Nv = 21760 image tokens, L text tokens; L = 194 by default, `L=4096 python tools/bench_xattn.py` for the
shipped eval padding) and of the decoder self-attention shape (8 heads x 32, 910 queries) on hipie_flash_attn."""
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
    B, Nv, H, hd = 8, 21760, 8, 256
    L = int(os.environ.get("L", "194"))
    dt = torch.float16 if os.environ.get("DT") == "f16" else torch.bfloat16
    g = torch.Generator().manual_seed(0)
    q = (torch.randn(B, Nv, H, hd, generator=g) * 0.06).to(dt).to(dev)
    vv = torch.randn(B, Nv, H, hd, generator=g).to(dt).to(dev)
    k = torch.randn(B, L, H, hd, generator=g).to(dt).to(dev)
    vl = torch.randn(B, L, H, hd, generator=g).to(dt).to(dev)
    mask = torch.ones(B, L, dtype=torch.bool, device=dev)
    t = bench(lambda: ops.bi_xattn(q, k, vv, vl, mask))
    fl = 2 * 4.0 * B * H * Nv * L * hd           # two passes of (QK^T + PV)
    print("bi_xattn L=%d %s: %.3f ms  %.0f TFLOP/s (both passes)" % (L, str(dt).split(".")[-1], t, fl / t / 1e9))
    N, h2, d2 = 910, 8, 32
    x = torch.randn(B, N, 3, h2, d2, generator=g).to(dt).to(dev)
    t = bench(lambda: ops.flash_attn(x[:, :, 0], x[:, :, 1], x[:, :, 2], d2 ** -0.5))
    print("flash_attn hd32 N=910: %.4f ms  %.1f TFLOP/s" % (t, 4.0 * B * h2 * N * N * d2 / t / 1e9))


if __name__ == "__main__":
    main()
