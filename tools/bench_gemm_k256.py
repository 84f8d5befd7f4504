#!/usr/bin/env python3
"""The K = 256 linears over all pyramid tokens (M = 174080 at bs 8) on hipie_gemm: the 256 x 256 tile kernel (HIPIE_GEMM_K256=0) against the
thin-K kernel of gemm_k256.hip (=1), same process, results checked against fp64 and against each other."""
import os
import sys

import torch

sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

# (M, N, a_is_f32, launches per step) -- profiles/r04_stage_times.txt
SHAPES = [(174080, 256, True, 34), (174080, 256, False, 12), (174080, 384, False, 12), (131072, 1024, True, 1), (174080, 2304, True, 0), (8200, 256, True, 0)]


def bench(fn, n=30):
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


def main():
    K = 256
    tot = {"0": 0.0, "1": 0.0}
    for M, N, f32, n in SHAPES:
        x = torch.randn(M, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        ws = ops.hl8_pack(w)
        a = x if f32 else ops.to_hl8(x)
        ref = (x[:4096].double() @ w.double().t() + b.double())
        outs, row = {}, []
        for mode in ("0", "1"):
            os.environ["HIPIE_GEMM_K256"] = mode
            outs[mode] = ops.gemm(a, ws, b, out_fmt=ops.F32, split=True)
            err = float((outs[mode][:4096].double() - ref).abs().max() / ref.abs().max())
            t = bench(lambda: ops.gemm(a, ws, b, out_fmt=ops.F32, split=True))
            tot[mode] += t * n
            row.append("K256=%s %.4f ms (err %.1e)" % (mode, t, err))
        d = float((outs["0"] - outs["1"]).abs().max() / outs["0"].abs().max())
        print("M=%6d N=%4d %s x %2d:  %s   thin vs tile %.1e" % (M, N, "f32" if f32 else "hl8", n, "  ".join(row), d), flush=True)
    os.environ.pop("HIPIE_GEMM_K256")
    print("sum over the step's launches: tile kernel %.2f ms, thin-K kernel %.2f ms" % (tot["0"], tot["1"]))


if __name__ == "__main__":
    main()
