#!/usr/bin/env python3
"""Bisect of the MSDA co-residency fault: variants of the fused kernel (tools/ubench/msda_study.hip) run on a side
stream while gemm_kernel<256> runs on the main stream; every launch is compared with the same variant run alone.  The DUMP variant records
what every lane loaded and computed, so a wrong group is traced to a load or to arithmetic.
    make -C tools/ubench -f Makefile.msda_study && python tools/msda_study.py [iterations]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hipie_amd import ops  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", os.environ.get("STUDY_LIB", "libmsda_study.so")))
lib.msda_study.restype = ctypes.c_int
lib.msda_study.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 8 + [ctypes.c_int] * 7 + [ctypes.c_void_p]
ITER = int(sys.argv[1]) if len(sys.argv) > 1 else 25


def rn(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(dev)


B, S, Q, M, L, P = 8, 21760, 300, 8, 4, 4
LP = L * P
shapes = torch.tensor([[128, 128], [64, 64], [32, 32], [16, 16]], device=dev)
lstart = torch.tensor([0, 16384, 20480, 21504], device=dev)
val = rn(B, S, 8, 32)
dref4 = (torch.rand(B, Q, 4, 4, generator=g) * 0.5 + 0.25).to(dev)
doff = rn(B, Q, 8, 4, 4, 2)
dlog = rn(B, Q, 8, 16)
uloc = torch.rand(B, Q, 8, 4, 4, 2, generator=g).to(dev)
bres = rn(174080, 256)
bw = ops.hl8_pack(rn(256, 256, scale=0.06)).to(dev)
t32 = rn(8, 300, 256)
NDBG = B * Q * M * LP * 20 + 4
FIELDS = ["o0", "o1", "o2", "o3", "w0", "w1", "w2", "w3", "x_raw", "y_raw", "r0", "r1", "r2", "r3", "logit", "wmax", "winv", "H", "W", "lstart"]


def launch(flags, out, dbg):
    off = uloc if flags & 4 else doff
    rc = lib.msda_study(flags, val.data_ptr(), shapes.data_ptr(), lstart.data_ptr(), dref4.data_ptr(), off.data_ptr(), dlog.data_ptr(),
                        out.data_ptr(), dbg.data_ptr() if dbg is not None else None, B, S, M, L, Q, P, 4,
                        torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc


BG = {
    "gemm<256> f32 rows": lambda: ops.gemm(bres, bw, None, split=True, out_fmt=ops.F32),
    "gemm_small": lambda: [ops.gemm(t32, bw, None, split=True, out_fmt=ops.F32) for _ in range(14)],
}
side = torch.cuda.Stream()
main = torch.cuda.current_stream()
NAMES = {0: "baseline", 1: "DUMP", 2: "NOSOFT", 4: "NOLOC", 6: "NOSOFT+NOLOC", 8: "NOTRANS", 16: "SHAPELDS", 32: "STATICLDS", 64: "BLOCKBAR",
         128: "TWICE", 256: "DPP", 48: "SHAPELDS+STATICLDS", 24: "NOTRANS+SHAPELDS", 129: "TWICE+DUMP", 18: "NOSOFT+SHAPELDS",
         20: "NOLOC+SHAPELDS", 8 + 16 + 32 + 256: "NOTRANS+SHAPELDS+STATICLDS+DPP", 512: "FLOATMASK", 1024: "ASMTIGHT", 2048: "ASMRAW (nop before select)",
         4096: "ASMWAR (nop after select)", 8192: "ASMNOVCC"}


def explain(flags, o, ref, dbg, dref):
    o, ref = o.view(B, Q, M, 32), ref.view(B, Q, M, 32)
    bad = torch.nonzero((o - ref).abs().amax(-1) > 0)
    print("    %d wrong groups; first %s" % (bad.shape[0], bad[:6].tolist()))
    if dbg is None:
        return
    d, r = dbg[:-4].view(B, Q, M, LP, 20), dref[:-4].view(B, Q, M, LP, 20)
    dd = (d.view(torch.int32) != r.view(torch.int32))
    gb = torch.nonzero(dd.flatten(3).any(-1))
    print("    dump: %d groups with differing dump words (of %d wrong outputs)" % (gb.shape[0], bad.shape[0]))
    for b, q, h in gb[:8].tolist():
        pts = torch.nonzero(dd[b, q, h].any(-1)).flatten().tolist()
        flds = sorted(set(torch.nonzero(dd[b, q, h])[:, 1].tolist()))
        print("      (b %d q %3d h %d): points %s fields %s" % (b, q, h, pts, [FIELDS[f] for f in flds]))
        i = pts[0]
        for f in flds:
            gv, wv = d[b, q, h, i, f], r[b, q, h, i, f]
            if f < 4:
                print("        pt %2d %-6s got %d want %d" % (i, FIELDS[f], int(gv.view(torch.int32)), int(wv.view(torch.int32))))
            else:
                print("        pt %2d %-6s got %.9g want %.9g" % (i, FIELDS[f], float(gv), float(wv)))
        # is the wrong raw input the correct value of ANOTHER place of the same input tensor?
        for f, name, src in ((14, "logit", dlog.view(-1)), (8, "x_raw", doff.view(-1)), (10, "r0", dref4.view(-1))):
            if f in flds:
                hit = torch.nonzero(src == d[b, q, h, i, f]).flatten().tolist()[:4]
                want = torch.nonzero(src == r[b, q, h, i, f]).flatten().tolist()[:4]
                print("        %s value found in the source tensor at flat index %s (the right one sits at %s)" % (name, hit, want))


def main_loop():
    order = [int(x) for x in os.environ.get("VARIANTS", "0,1,129,2,4,6,8,16,32,64,256,24,48,%d" % (8 + 16 + 32 + 256)).split(",")]
    for bname, bg in BG.items():
        print("=== background: %s" % bname, flush=True)
        for flags in order:
            want_dbg = bool(flags & 1) or bool(flags & 128)
            ref = torch.empty(B, Q, M * 32, device=dev)
            dref = torch.zeros(NDBG, device=dev) if want_dbg else None
            launch(flags, ref, dref)
            torch.cuda.synchronize()
            solo_bad = 0
            for _ in range(3):
                o2 = torch.empty_like(ref)
                d2 = torch.zeros(NDBG, device=dev) if want_dbg else None
                launch(flags, o2, d2)
                torch.cuda.synchronize()
                solo_bad += int(not torch.equal(o2, ref))
            bad, total, twice = 0, 0, 0
            shown = 0
            for it in range(ITER):
                outs = [torch.empty_like(ref) for _ in range(4)]
                dbgs = [torch.zeros(NDBG, device=dev) if want_dbg else None for _ in range(4)]
                torch.cuda.synchronize()
                side.wait_stream(main)
                for _ in range(3):
                    bg()
                with torch.cuda.stream(side):
                    for o, d in zip(outs, dbgs):
                        launch(flags, o, d)
                main.wait_stream(side)
                torch.cuda.synchronize()
                for o, d in zip(outs, dbgs):
                    total += 1
                    if d is not None and flags & 128:
                        twice += int(d[-4:].view(torch.int32)[0])
                    if not torch.equal(o, ref):
                        bad += 1
                        if shown < 2:
                            shown += 1
                            explain(flags, o, ref, d if flags & 1 else None, dref)
            print("variant %4d %-34s: %3d / %3d launches differ (alone: %d / 3)%s" % (
                flags, NAMES.get(flags, "?"), bad, total, solo_bad, "  TWICE: %d differing record words" % twice if flags & 128 else ""), flush=True)


def product_table():
    """the product kernel beside the kernels of the step (which backgrounds trigger it?)"""
    bx = ops.to_hl8(bres)
    vx = ops.to_hl8(rn(32768, 1280))
    vw = ops.hl8_pack(rn(1280, 1280, scale=0.03)).to(dev)
    bw384 = ops.hl8_pack(rn(384, 256, scale=0.06)).to(dev)
    lin1 = torch.nn.Linear(256, 2048).to(dev)
    lin2 = torch.nn.Linear(2048, 256).to(dev)
    emb = rn(8, 300, 256)
    feats = rn(8, 256, 256, 256)
    lnw, lnb = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    bgs = {
        "none": lambda: None,
        "gemm<256> HL8 rows": lambda: ops.gemm(bx, bw, None, split=True, out_fmt=ops.F32),
        "gemm<256> f32 rows": BG["gemm<256> f32 rows"],
        "gemm_small": BG["gemm_small"],
        "gemm k256 thin (N 384)": lambda: ops.gemm(bx, bw384, None, split=True, out_fmt=ops.F32),
        "gemm<320> ViT": lambda: ops.gemm(vx, vw, None, split=True, out_fmt=ops.F32),
        "ffn_fused": lambda: ops.ffn_fused(bx.view(8, 21760, 512), lin1, lin2),
        "layernorm dec": lambda: ops.add_layernorm_dec(bres, bres, lnw, lnb, 1e-5, "hl8", want16=True),
        "to_hl8": lambda: ops.to_hl8(bres),
        "mask_einsum": lambda: ops.mask_einsum(emb, feats, precision=1),
        "torch mm f32": lambda: torch.mm(bres, bres[:256].t()),
        "torch add": lambda: bres + 1.0,
    }
    fn = lambda: ops.msda_fused(val, shapes, lstart, dref4, doff, dlog)
    ref = fn().clone()
    torch.cuda.synchronize()
    print("=== product hipie_msda_fused (decoder form) beside the step's kernels", flush=True)
    for bname, bg in bgs.items():
        bad = 0
        for it in range(10):
            side.wait_stream(main)
            for _ in range(3):
                bg()
            with torch.cuda.stream(side):
                outs = [fn() for _ in range(4)]
            main.wait_stream(side)
            torch.cuda.synchronize()
            bad += sum(int(not torch.equal(o, ref)) for o in outs)
        print("  beside %-24s: %2d / 40 differ" % (bname, bad), flush=True)


if __name__ == "__main__":
    if os.environ.get("TABLE", "1") == "1":
        product_table()
    main_loop()
