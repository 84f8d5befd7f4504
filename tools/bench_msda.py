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
    """hipie_msda_backward at the encoder's training geometry (B = 2 images, every one of the 21760 tokens a query), fp32.  Two location
    distributions: `near` = the query's own reference point + N(0, SIGMA px) offsets (what the encoder produces; SIGMA = 1.5 by default) and
    `uniform` = anywhere on the map (the worst case for the coarse levels: every query of an image meets on 256 pixels).  Both forms of the
    operator: the gather form (the one ops.ms_deform_attn_backward runs for fp32, D = 32) and the atomic kernel."""
    dev = "cuda"
    B, M, D, L, P = 2, 8, 32, 4, 4
    shapes = [(128, 128), (64, 64), (32, 32), (16, 16)]
    S = sum(h * w for h, w in shapes)
    g = torch.Generator().manual_seed(0)
    value = torch.randn(B, S, M, D, generator=g).to(dev)
    attn = torch.softmax(torch.randn(B, S, M, L * P, generator=g), -1).view(B, S, M, L, P).to(dev)
    gout = torch.randn(B, S, M * D, generator=g).to(dev)
    ss = torch.tensor(shapes, device=dev)
    ls = torch.cat((ss.new_zeros((1,)), ss.prod(1).cumsum(0)[:-1]))
    ref = encoder_reference_points(shapes, torch.ones(B, L, 2), "cpu")                    # (B, S, L, 2)
    wh = torch.tensor([[w, h] for h, w in shapes], dtype=torch.float32)
    sigma = float(os.environ.get("SIGMA", "1.5"))
    near = ref[:, :, None, :, None, :] + sigma * torch.randn(B, S, M, L, P, 2, generator=g) / wh[None, None, None, :, None, :]
    locs = {"near": near.to(dev), "uniform": torch.rand(B, S, M, L, P, 2, generator=g).to(dev)}
    n = 10
    pts = B * S * M * L * P
    from hipie_amd import _lib
    lib = _lib.load()

    def atomic_form(loc):
        gv, gl, ga = torch.empty_like(value), torch.empty_like(loc), torch.empty_like(attn)
        rc = lib.hipie_msda_backward(value.data_ptr(), ss.data_ptr(), ls.data_ptr(), loc.data_ptr(), attn.data_ptr(), gout.data_ptr(), gv.data_ptr(),
                                     gl.data_ptr(), ga.data_ptr(), B, S, M, D, L, S, P, 0, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "hipie_msda_backward")
        return gv, gl, ga

    for name, loc in locs.items():
        res = {}
        for form, fn in (("gather (hipie_msda_backward_ws)", lambda: ops.ms_deform_attn_backward(value, ss, ls, loc, attn, gout)),
                         ("atomic (hipie_msda_backward)", lambda: atomic_form(loc))):
            for _ in range(3):
                res[form] = fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t) / n * 1e3
            print("msda_backward B=%d Lq=%d loc=%s %s: %.3f ms, %.1f G corners of %d floats / s" % (B, S, name, form, ms, pts * 4 / ms / 1e6, D))
        a_, b_ = list(res.values())
        print("   gather vs atomic, max |diff| / max |ref|:", ["%.1e" % float((x - y).abs().max() / y.abs().max()) for x, y in zip(a_, b_)])
        f = ops.ms_deform_attn_forward(value, ss, ls, loc, attn)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            ops.ms_deform_attn_forward(value, ss, ls, loc, attn)
        torch.cuda.synchronize()
        print("msda_forward (unfused op, same inputs) loc=%s: %.3f ms" % (name, (time.perf_counter() - t) / n * 1e3))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "bwd":
        backward()
    else:
        main()
