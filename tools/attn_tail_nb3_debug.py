"""Debugging aid for the 16-row tail of hipie_vit_attn_split on the 96-slot (NB = 3) instance: 84 x 84 grid, head dim 80.
Run with HIPIE_LIB_PATH=tools/ubench/_build/libhipie_nb3.so (a build with -DHIPIE_VS_TAIL_MAXNB=3).  Prints the error against the
materialised fp64 formulation by d-column block, by query half of a 32-query wave tile, and with V restricted to one 32-key block of a row."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from hipie_amd import ops  # noqa: E402

DEV = "cuda"


def run(gh, gw, heads, hd, vmask=None, seed=3):
    gen = torch.Generator().manual_seed(seed)
    B, N, C = 1, gh * gw, heads * hd
    scale = hd ** -0.5
    qkv = torch.randn(B, N, 3 * C, generator=gen) * 1.0
    if vmask is not None:
        v = qkv[..., 2 * C:].view(B, gh, gw, C)
        keep = torch.zeros(gw, dtype=torch.bool)
        keep[vmask[0]:vmask[1]] = True
        v[:, :, ~keep] = 0
    th32, tw32 = torch.randn(2 * gh - 1, hd, generator=gen) * 0.2, torch.randn(2 * gw - 1, hd, generator=gen) * 0.2
    f = qkv.clone()
    f[..., :C] *= scale * ops.LOG2E
    q, k, v = (t.to(DEV).double().view(B, heads, N, hd) for t in qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4))
    ih = torch.arange(gh, device=DEV)[:, None] - torch.arange(gh, device=DEV)[None, :] + gh - 1
    iw = torch.arange(gw, device=DEV)[:, None] - torch.arange(gw, device=DEV)[None, :] + gw - 1
    Rh, Rw = th32.to(DEV).double()[ih], tw32.to(DEV).double()[iw]
    rq = q.reshape(B, heads, gh, gw, hd)
    attn = (q * scale) @ k.transpose(-2, -1)
    attn = (attn.view(B, heads, gh, gw, gh, gw) + torch.einsum("bmhwc,hkc->bmhwk", rq, Rh)[..., :, None]
            + torch.einsum("bmhwc,wkc->bmhwk", rq, Rw)[..., None, :]).view(B, heads, N, N)
    want = (attn.softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B, N, heads, hd).float().cpu()
    got = ops.hl8_unpack(ops.vit_attn_split(ops.hl8_pack(f).to(DEV), ops.hl8_pack(th32 / scale).to(DEV), ops.hl8_pack(tw32 / scale).to(DEV),
                                            (gh, gw), heads)).cpu().view(B, N, heads, hd)
    err = (got - want).abs() / want.abs().max()
    out = {"all": float(err.max())}
    for name, sl in (("d0-31", slice(0, 32)), ("d32-63", slice(32, 64)), ("d64-79", slice(64, 80))):
        out[name] = float(err[..., sl].max())
    qi = torch.arange(N)
    out["tail q%32<16"] = float(err[:, (qi % 32) < 16][..., 64:].max())
    out["tail q%32>=16"] = float(err[:, (qi % 32) >= 16][..., 64:].max())
    out["finite"] = bool(torch.isfinite(got).all())
    # is it a per-row scale?  least-squares factor per (query, head) row and what is left after dividing it out
    g2, w2 = got.view(N, heads, hd).double(), want.view(N, heads, hd).double()
    fac = (g2 * w2).sum(-1) / (w2 * w2).sum(-1)
    resid = ((g2 / fac[..., None] - w2).abs().amax(-1) / w2.abs().amax(-1))
    out["row factor min/med/max"] = [float(fac.min()), float(fac.median()), float(fac.max())]
    out["residual after rescale (max)"] = float(resid.max())
    bad = (fac - 1).abs() > 1e-3
    out["rows off"] = int(bad.sum())
    if bad.any():
        idx = torch.nonzero(bad[:, 0])[:, 0]
        out["first bad queries (head 0)"] = idx[:12].tolist()
        out["bad by qx%32<16 / >=16"] = [int(((idx % gw) % 32 < 16).sum()), int(((idx % gw) % 32 >= 16).sum())]
        out["bad by q%32<16 / >=16"] = [int((idx % 32 < 16).sum()), int((idx % 32 >= 16).sum())]
        out["factors of first bad"] = [round(float(x), 4) for x in fac[idx[:12], 0]]
    return out


if __name__ == "__main__":
    print("lib:", os.environ.get("HIPIE_LIB_PATH", "(in-tree)"))
    for gh, gw in ((4, 96), (4, 84), (40, 96)):
        print(gh, gw, run(gh, gw, 2, 80))
