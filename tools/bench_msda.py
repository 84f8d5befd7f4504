#!/usr/bin/env python3
"""micro-benchmark of the deformable-attention sampling kernels on the encoder geometry of the bench workload
(B = 8, pyramid 128/64/32/16, 8 heads x 32, 4 levels x 4 points, bf16 value, bf16 strided offsets/logits)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hipie_amd import ops  # noqa: E402
from hipie_amd.modeling.transformer import encoder_reference_points  # noqa: E402


def main():
    dev = "cuda"
    B, M, D, L, P = 8, 8, 32, 4, 4
    shapes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(0)
    value = torch.randn(B, S, M, D, generator=g).bfloat16().to(dev)
    proj = torch.randn(B, S, M * L * P * 3, generator=g)
    proj[..., :M * L * P * 2] *= float(os.environ.get("SIGMA", "1.5"))
    proj = proj.bfloat16().to(dev)
    off = proj[..., :M * L * P * 2].unflatten(-1, (M, L, P, 2))
    lg = proj[..., M * L * P * 2:].unflatten(-1, (M, L * P))
    ref = encoder_reference_points(shapes, torch.ones(B, L, 2), "cpu").to(dev)
    ss = torch.tensor(shapes, device=dev)
    ls = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))

    def bench(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    alg = B * (S * M * D * 2 * 2 + S * M * L * P * 3 * 2 + S * L * 2 * 4)         # value in + out, offsets + logits, refs
    t = bench(lambda: ops.msda_fused(value, ss, ls, ref, off, lg))
    print("msda_fused: %.3f ms  %.2f TB/s algorithmic" % (t, alg / t / 1e9))
    # the split3 policy's call: fp32 value / offsets / logits, every group -> workgroup map of the kernel (HIPIE_MSDA_MAP)
    v32, o32, l32 = value.float(), off.float().contiguous(), lg.float().contiguous()
    alg32 = B * (S * M * D * 4 * 2 + S * M * L * P * 3 * 4 + S * L * 2 * 4)
    outs = {}
    for mp in ("0", "1", "2"):
        os.environ["HIPIE_MSDA_MAP"] = mp
        t = bench(lambda: ops.msda_fused(v32, ss, ls, ref, o32, l32))
        outs[mp] = ops.msda_fused(v32, ss, ls, ref, o32, l32)
        print("msda_fused f32 map %s: %.3f ms  %.2f TB/s algorithmic  (sigma %s)" % (mp, t, alg32 / t / 1e9, os.environ.get("SIGMA", "1.5")))
    os.environ.pop("HIPIE_MSDA_MAP")
    print("maps bit-identical:", torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["2"]))


def backward():
    """hipie_msda_backward at the encoder's training geometry (B = 2 images, every one of the 21760 tokens a query), fp32."""
    dev = "cuda"
    B, M, D, L, P = 2, 8, 32, 4, 4
    shapes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(0)
    value = torch.randn(B, S, M, D, generator=g).to(dev)
    loc = torch.rand(B, S, M, L, P, 2, generator=g).to(dev)
    attn = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).to(dev)
    gout = torch.randn(B, S, M * D, generator=g).to(dev)
    ss = torch.tensor(shapes, device=dev)
    ls = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    for _ in range(3):
        ops.ms_deform_attn_backward(value, ss, ls, loc, attn, gout)
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for _ in range(n):
        ops.ms_deform_attn_backward(value, ss, ls, loc, attn, gout)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / n * 1e3
    pts = B * S * M * L * P
    print("msda_backward B=%d Lq=%d: %.3f ms (incl. the grad_value memset), %.1f G corner atomics of %d floats / s, corner traffic %.2f TB/s"
          % (B, S, ms, pts * 4 / ms / 1e6, D, pts * 4 * D * 4 * 2 / ms / 1e9))
    f = ops.ms_deform_attn_forward(value, ss, ls, loc, attn)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        ops.ms_deform_attn_forward(value, ss, ls, loc, attn)
    torch.cuda.synchronize()
    print("msda_forward (unfused op, same inputs): %.3f ms" % ((time.perf_counter() - t) / n * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":
        backward()
    else:
        main()
