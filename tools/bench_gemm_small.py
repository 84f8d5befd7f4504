#!/usr/bin/env python3
"""The small linears of the step (decoders, BERT, heads) on hipie_gemm: run once with HIPIE_GEMM_SMALL=0 (256-row tiles only) and once
with the default (64 x 128 tile kernel for problems that fill < 3/4 of the CUs).  Prints ms per launch and checks the result against fp64."""
import os
import sys

import torch

sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

# (M, K, N, a_is_f32, launches per step) -- profiles/r03_stage_times.txt
SHAPES = [(2400, 256, 256, True, 64), (7280, 256, 256, True, 40), (2400, 2048, 256, True, 9), (7280, 2048, 256, True, 6), (2400, 512, 256, True, 9),
          (2400, 256, 2048, True, 9), (7280, 256, 2048, True, 6), (2400, 256, 8, True, 10), (1552, 3072, 768, False, 12), (1552, 768, 3072, False, 12),
          (1552, 768, 2304, False, 12), (1552, 768, 768, True, 12), (8192, 1280, 256, True, 2), (32768, 1280, 256, True, 2), (7280, 256, 176, True, 1)]


def bench(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# the K = 256 projections over all pyramid tokens (tile kernel by default; HIPIE_GEMM_SMALL_MAXTILES=1000000 forces the 64 x 128 tile kernel)
BIG = [(174080, 256, 256, True, 34), (174080, 256, 256, False, 12), (174080, 256, 384, False, 12), (131072, 256, 1024, True, 1)]


def main():
    tot = 0.0
    for M, K, N, f32, n in (BIG if len(sys.argv) > 1 and sys.argv[1] == "big" else SHAPES):
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        ws = ops.hl8_pack(w)
        a = x if f32 else ops.to_hl8(x)
        out = ops.gemm(a, ws, b, out_fmt=ops.F32, split=True)
        ref = (x.double() @ w.double().t() + b.double())
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        t = bench(lambda: ops.gemm(a, ws, b, out_fmt=ops.F32, split=True), 20 if M > 100000 else 50)
        tot += t * n
        print("M=%6d K=%4d N=%4d %s  %.4f ms x %2d   err %.1e" % (M, K, N, "f32" if f32 else "hl8", t, n, err), flush=True)
    print("HIPIE_GEMM_SMALL=%s: sum over the step's launches %.2f ms" % (os.environ.get("HIPIE_GEMM_SMALL", "1"), tot))


if __name__ == "__main__":
    main()
