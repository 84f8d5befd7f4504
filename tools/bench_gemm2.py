#!/usr/bin/env python3
"""hipie_gemm (plain fp16 / split HL8) vs the library GEMMs on the ViT-H linears (M = batch * 4096 tokens) and the head shapes.
Prints ms, algorithmic TFLOP/s (2MNK / t) and, for the split form, the MFMA-issue rate (3x)."""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from hipie_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
SHAPES = [("qkv", M, 1280, 3840), ("proj", M, 1280, 1280), ("fc1", M, 1280, 5120), ("fc2", M, 5120, 1280),
          ("enc_ffn1", 174080, 256, 2048), ("enc_ffn2", 174080, 2048, 256), ("dec", 7280, 256, 256)]


def bench(fn, n=10):
    for _ in range(2):
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
    for name, m, K, N in SHAPES:
        x = torch.randn(m, K, device="cuda")
        w = torch.randn(N, K, device="cuda") * K ** -0.5
        b = torch.randn(N, device="cuda")
        x16, w16, b16 = x.half(), w.half(), b.half()
        xs, ws = ops.to_hl8(x), ops.hl8_pack(w)
        gf = 2.0 * m * K * N / 1e9
        t_lib16 = bench(lambda: F.linear(x16, w16, b16))
        t_lib32 = bench(lambda: F.linear(x, w, b), n=3)
        t_p = bench(lambda: ops.gemm(x16, w16, b, out_fmt=ops.F16, split=False))
        t_s = bench(lambda: ops.gemm(xs, ws, b, out_fmt=ops.F32, split=True))
        t_sg = bench(lambda: ops.gemm(xs, ws, b, out_fmt=ops.HL8, act=ops.ACT_GELU, split=True))
        print("%-9s M=%6d K=%4d N=%4d | lib fp16 %.3f ms %5.0f TF | lib fp32 %.3f ms %4.0f TF | hipie fp16 %.3f ms %5.0f TF | "
              "hipie split %.3f ms %4.0f TF (MFMA %5.0f) | split+gelu->hl8 %.3f ms" %
              (name, m, K, N, t_lib16, gf / t_lib16, t_lib32, gf / t_lib32, t_p, gf / t_p, t_s, gf / t_s, 3 * gf / t_s, t_sg), flush=True)
        del x, w, x16, w16, xs, ws


def variants():
    import os
    x = torch.randn(M, 1280, device="cuda")
    b = torch.randn(3840, device="cuda")
    xs, ws = ops.to_hl8(x), ops.hl8_pack(torch.randn(3840, 1280, device="cuda") * 0.03)
    x2 = ops.to_hl8(torch.randn(M, 5120, device="cuda"))
    w2 = ops.hl8_pack(torch.randn(1280, 5120, device="cuda") * 0.02)
    for v in ("0", "1", "0"):       # 1 = no epilogue (needs a library built with make EXTRA=-DHIPIE_GEMM_VARIANTS)
        os.environ["HIPIE_GEMM_VARIANT"] = v
        t = bench(lambda: ops.gemm(xs, ws, b, out_fmt=ops.F32, split=True), n=20)
        th = bench(lambda: ops.gemm(xs, ws, b, out_fmt=ops.HL8, split=True), n=20)
        t2 = bench(lambda: ops.gemm(x2, w2, None, out_fmt=ops.F32, split=True), n=20)
        print("variant %s: qkv split %.3f ms (MFMA %.0f TF), HL8 out %.3f ms   fc2 split %.3f ms (MFMA %.0f TF)" %
              (v, t, 6.0 * M * 1280 * 3840 / t / 1e9, th, t2, 6.0 * M * 1280 * 5120 / t2 / 1e9), flush=True)


def epilogue_only():
    """variant 3 = the epilogue alone (no main loop): whole qkv grid (1536 tiles = 6 rounds of 256 CUs), 32 tiles (4 CUs per XCD: the
    uncontended per-CU epilogue latency) and 1 tile (launch overhead)."""
    import os
    b = torch.randn(3840, device="cuda")
    ws = ops.hl8_pack(torch.randn(3840, 1280, device="cuda") * 0.03)
    for m, n in ((M, 3840), (M, 1280), (65536, 320), (8192, 320), (256, 320)):
        xs = ops.to_hl8(torch.randn(m, 1280, device="cuda"))
        for v in ("0", "3"):
            os.environ["HIPIE_GEMM_VARIANT"] = v
            t = bench(lambda: ops.gemm(xs, ws[:n], b[:n], out_fmt=ops.F32, split=True), n=20)
            th = bench(lambda: ops.gemm(xs, ws[:n], b[:n], out_fmt=ops.HL8, split=True), n=20)
            tg = bench(lambda: ops.gemm(xs, ws[:n], b[:n], out_fmt=ops.HL8, act=ops.ACT_GELU, split=True), n=20)
            print("M=%6d N=%4d (%4d tiles) variant %s: fp32 out %.4f ms, HL8 out %.4f ms, GELU+HL8 out %.4f ms; out bytes %.0f MB" %
                  (m, n, (m // 256) * (n // 320), v, t, th, tg, m * n * 4 / 1e6), flush=True)
    os.environ["HIPIE_GEMM_VARIANT"] = "0"


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "variants":
        variants()
    elif len(sys.argv) > 2 and sys.argv[2] == "epilogue":
        epilogue_only()
    else:
        main()
