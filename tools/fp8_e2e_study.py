#!/usr/bin/env python3
"""End-to-end emulation of the split GEMM with its two CROSS terms on block-scaled e4m3 operands (docs/next_round.md: 2 fp16-equivalent
MFMAs per product instead of 3).  ops.gemm is replaced, for the launches selected by `which`, by a torch emulation at ACCUMULATOR level
    acc = A_hi . W_hi^T  +  q8(A_hi) . q8(W_lo)^T  +  q8(A_lo) . q8(W_hi)^T        (fp32 accumulation)
followed by hipie_gemm's own epilogue arithmetic (alpha, bias, GELU / ReLU / QuickGELU, residual, output scale, fp32 / fp16 / HL8 output, row
maps); everything else of the step runs on the product kernels.  Prints the a22 errors against the reference goldens on the gate fixtures.
    python tools/fp8_e2e_study.py [vit|all] [e2e_tiny,e2e_deep,e2e_full_c80]
`vit`: only the ViT linears (K >= 768); `all`: every split GEMM that goes through ops.gemm (the fused FFN / attention kernels keep three products);
`qkv` | `proj` | `fc1` | `fc2` or sums like `qkv+proj`: those ViT-H linears alone."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402

torch.set_grad_enabled(False)
import bench  # noqa: E402
import test_gpu_e2e as T  # noqa: E402
from hipie_amd import ops  # noqa: E402
from hipie_amd.config import Precision  # noqa: E402
from util import rel_err  # noqa: E402

REAL = ops.gemm
MODE = {"which": "vit", "on": True, "count": 0}


def split(x):
    hi = x.half().float()
    return hi, (x - hi).half().float()


def q8(x, block=32):
    shp = x.shape
    xb = x.reshape(-1, block)
    amax = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(torch.floor(torch.log2(448.0 / amax)))
    return ((xb * scale).to(torch.float8_e4m3fn).float() / scale).reshape(shp)


def emu_gemm(a, w, bias=None, resid=None, out_fmt=ops.F32, act=ops.ACT_NONE, alpha=1.0, oscale=1.0, split_=None, out=None, tag="gemm", out_row=None,
             out_rows=None, a_row=None, **kw):
    split_ = kw.pop("split", split_)
    K = w.shape[1] // 2 if split_ else w.shape[1]
    fam = {(1280, 3840): "qkv", (1280, 1280): "proj", (1280, 5120): "fc1", (5120, 1280): "fc2"}.get((K, w.shape[0]))
    sel = MODE["which"] == "all" or (MODE["which"] == "vit" and K >= 768) or (fam is not None and fam in MODE["which"].split("+"))
    if not (MODE["on"] and split_ and sel):
        return REAL(a, w, bias, resid, out_fmt=out_fmt, act=act, alpha=alpha, oscale=oscale, split=split_, out=out, tag=tag, out_row=out_row,
                    out_rows=out_rows, a_row=a_row)
    MODE["count"] += 1
    N = w.shape[0]
    W = ops.hl8_unpack(w)
    A = a.reshape(-1, K).float() if a.dtype == torch.float32 else ops.hl8_unpack(a.reshape(-1, 2 * K))
    lead = a.shape[:-1]
    if a_row is not None:
        A = A[a_row.long()]
        lead = (A.shape[0],)
    ah, al = split(A)
    wh, wl = split(W)
    acc = ah @ wh.t() + q8(ah) @ q8(wl).t() + q8(al) @ q8(wh).t()
    y = alpha * acc + (0 if bias is None else bias)
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = torch.relu(y)
    elif act == 3:
        y = y * torch.sigmoid(1.702 * y)
    M = y.shape[0]
    rows = None if out_row is None else out_row.long()
    if resid is not None:
        r2 = resid.reshape(-1, N)
        y = y + (r2[:M] if rows is None else r2[rows.clamp_min(0)])
    y = y * oscale

    def fmt(t):
        return t if out_fmt == ops.F32 else (t.half() if out_fmt == ops.F16 else ops.to_hl8(t.contiguous()))
    if out is None and rows is None:
        return fmt(y).view(*lead, -1)
    if out is None:
        width = N if out_fmt != ops.HL8 else 2 * N
        out = torch.zeros(int(out_rows), width, dtype=torch.float32 if out_fmt == ops.F32 else torch.float16, device=a.device)
    o2 = out.reshape(-1, out.shape[-1])
    v = fmt(y)
    if rows is None:
        o2[:M] = v
    else:
        keep = rows >= 0
        o2[rows[keep]] = v[keep]
    return out


def main():
    MODE["which"] = sys.argv[1] if len(sys.argv) > 1 else "vit"
    fixtures = (sys.argv[2] if len(sys.argv) > 2 else "e2e_tiny,e2e_deep,e2e_full_c80").split(",")
    ops.gemm = emu_gemm
    for fx in fixtures:
        g, model = T.build(Precision.split3(), fx)
        task = "detection"
        if "bench_inputs" in g.meta:
            bi = g.meta["bench_inputs"]
            batch = bench.synth_batch(None, 1, bi["size"], bi["n_classes"], bi["L"], "cpu", seed=bi["seed"], task=task)
        else:
            batch = T.inputs(g, task)
        model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
        for on in (False, True):
            MODE["on"], MODE["count"] = on, 0
            out = model.forward_raw(batch)
            errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in T.KEYS}
            print("%-13s %-26s (%3d emulated launches) max %.1e | " % (fx, "fp8 cross terms [%s]" % MODE["which"] if on else "three fp16 products", MODE["count"],
                                                                       max(errs.values())) + " ".join("%s=%.1e" % (k.replace("pred_", ""), v) for k, v in errs.items()), flush=True)
        del model
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
