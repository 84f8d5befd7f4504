#!/usr/bin/env python3
"""hipie_gemm_ln (output_proj + residual + LayerNorm in one launch) against the two launches it replaces, at the encoder's size
(8 x 21760 pyramid tokens, K = N = 256), and the batched decoder value projection (N = 6 / 9 x 256 on the thin-K kernel) against one
256-column GEMM per layer:  python tools/bench_gemm_ln.py"""
import sys
import torch
sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

torch.set_grad_enabled(False)


class Owner:
    pass


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


M, K = 8 * 21760, 256
x = torch.randn(M, K, device="cuda")
w = torch.randn(256, K, device="cuda") * K ** -0.5
b = torch.randn(256, device="cuda")
resid = torch.randn(M, 256, device="cuda")
g, be = torch.randn(256, device="cuda"), torch.randn(256, device="cuda")
wh = ops.hl8_pack(w)
own = Owner()
for name, a, hl8 in (("fp32 A", x, False), ("HL8 A", ops.to_hl8(x), True)):
    def two():
        p = ops.gemm(a, wh, b, split=True)
        return ops.add_layernorm_dec(resid, p, g, be, 1e-5, "hl8", want16=True)

    def one():
        return ops.split_linear_ln(a, own, "w", w, b, resid, g, be, 1e-5, x_hl8=hl8)

    t_gemm = timed(lambda: ops.gemm(a, wh, b, split=True))
    t2, t1 = timed(two), timed(one)
    err = (two()[0] - one()[0]).abs().max().item()
    print("gemm_ln %-7s M=%d: gemm %.3f ms, gemm + add_layernorm_dec %.3f ms, fused %.3f ms (max |diff| %.2e)" % (name, M, t_gemm, t2, t1, err))

for layers in (6, 9):
    ws = [torch.randn(256, K, device="cuda") * K ** -0.5 for _ in range(layers)]
    whs = [ops.hl8_pack(v) for v in ws]
    wcat = ops.hl8_pack(torch.cat(ws, 0))
    bs = [torch.randn(256, device="cuda") for _ in range(layers)]
    bcat = torch.cat(bs)
    t_each = timed(lambda: [ops.gemm(x, whs[i], bs[i], split=True) for i in range(layers)])
    t_cat = timed(lambda: ops.gemm(x, wcat, bcat, split=True))
    print("decoder values, %d layers: one GEMM per layer %.3f ms, one batched GEMM (N = %d) %.3f ms" % (layers, t_each, layers * 256, t_cat))
