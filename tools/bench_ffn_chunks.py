#!/usr/bin/env python3
"""Does the encoder FFN run faster image by image?  linear1 (256 -> 2048, ReLU, HL8 out) writes a 1.4 GB hidden tensor for the bs-8 batch
(178 MB per image: inside the 256 MB Infinity Cache) that linear2 reads straight back.  Same split GEMMs, full batch vs chunks of one image."""
import sys
import torch
sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

B, Nv, C, Hd = 8, 21760, 256, 2048
x = ops.to_hl8(torch.randn(B * Nv, C, device="cuda"))
w1, b1 = ops.hl8_pack(torch.randn(Hd, C) * C ** -0.5).cuda(), torch.randn(Hd, device="cuda")
w2, b2 = ops.hl8_pack(torch.randn(C, Hd) * Hd ** -0.5).cuda(), torch.randn(C, device="cuda")
res = torch.randn(B * Nv, C, device="cuda")


def full():
    h = ops.gemm(x, w1, b1, out_fmt=ops.HL8, act=ops.ACT_RELU, split=True)
    return ops.gemm(h, w2, b2, resid=res, split=True)


def chunked(nc):
    rows = B * Nv // nc
    out = torch.empty(B * Nv, C, device="cuda")
    for i in range(nc):
        sl = slice(i * rows, (i + 1) * rows)
        h = ops.gemm(x[sl], w1, b1, out_fmt=ops.HL8, act=ops.ACT_RELU, split=True)
        ops.gemm(h, w2, b2, resid=res[sl], split=True, out=out[sl])
    return out


def bench(fn, n=10):
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


ref = full()
for nc in (4, 8, 16):
    assert torch.equal(chunked(nc), ref)
print("encoder FFN, bs 8 (174080 tokens): full batch %.3f ms" % bench(full))
for nc in (4, 8, 16):
    print("   %2d chunks of %6d tokens (hidden %4d MB each): %.3f ms" % (nc, B * Nv // nc, B * Nv // nc * Hd * 4 >> 20, bench(lambda: chunked(nc))))
