#!/usr/bin/env python3
"""CPU study of operand-rounding policies at the SHIPPED depths (tests/golden/e2e_deep.npz): which operands of which stage may be
16 bit, which need a hi + lo split, for the a22 outputs to stay within 1e-3 of the reference.

Runs the oracle (fp32 torch on the CPU) with its matrix products wrapped so that each operand is rounded the way a kernel
policy would round it (products and sums themselves stay fp32 = MFMA fp32 accumulation):
    f   exact fp32 operand
    h   one fp16 value                       (plain 16-bit operand)
    s   fp16 hi + fp16 lo (22 mantissa bits) (split operand: two MFMA passes for this side)
    b   one bf16 value
A policy maps a region (vit_lin, vit_attn_qk, vit_attn_pv, bert, head_lin, head_attn, conv, einsum, ...) to (lhs, rhs) modes.
Test infrastructure / design tool only: nothing under hipie_amd/ imports it.

    python tests/study/prec_sim.py [policy ...]
"""
import contextlib
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _synth  # noqa: E402
from util import Golden, rel_err  # noqa: E402
import oracle.model as om  # noqa: E402
import oracle.ops as oo  # noqa: E402

torch.set_grad_enabled(False)
KEYS = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino",
        "pred_logits_maskdino", "pred_boxes_maskdino"]


def rnd(x, m):
    if m == "f" or not x.is_floating_point():
        return x
    if m == "h":
        return x.half().float()
    if m == "b":
        return x.bfloat16().float()
    if m == "s":
        hi = x.half().float()
        return hi + (x - hi).half().float()
    raise KeyError(m)


class Sim:
    region = "other"
    policy = {}
    stream = {}      # region -> mode of the tensors a stage hands to the next one ("act" storage)

    @classmethod
    def modes(cls):
        return cls.policy.get(cls.region, cls.policy.get("default", ("f", "f")))


@contextlib.contextmanager
def region(name):
    old = Sim.region
    Sim.region = name
    try:
        yield
    finally:
        Sim.region = old


_lin, _conv, _convt, _matmul, _bmm, _einsum = F.linear, F.conv2d, F.conv_transpose2d, torch.matmul, torch.bmm, torch.einsum
_tmatmul = torch.Tensor.__matmul__


def p_linear(x, w, b=None):
    if Sim.region == "vit_lin":          # the four ViT linears by weight shape: qkv (3C, C), proj (C, C), fc1 (4C, C), fc2 (C, 4C)
        n, k = w.shape
        sub = "vit_qkv" if n == 3 * k else "vit_proj" if n == k else "vit_fc1" if n == 4 * k else "vit_fc2"
        a, c = Sim.policy.get(sub, Sim.modes())
    else:
        a, c = Sim.modes()
    return _lin(rnd(x, a), rnd(w, c), b)


def p_conv(x, w, b=None, **kw):
    a, c = Sim.policy.get("conv", Sim.modes()) if w.shape[-1] > 1 and Sim.region == "head" else Sim.modes()
    return _conv(rnd(x, a), rnd(w, c), b, **kw)


def p_convt(x, w, b=None, **kw):
    a, c = Sim.modes()
    return _convt(rnd(x, a), rnd(w, c), b, **kw)


def p_matmul(x, y):
    a, c = Sim.modes()
    if Sim.region in ("bert", "mha"):     # the attention products of BERT / nn.MultiheadAttention (their linears go through F.linear)
        a, c = Sim.policy.get(Sim.region + "_attn", (a, c))
        r = _matmul(rnd(x, a), rnd(y, c))
        if x.shape[-1] == x.shape[-2] and x.dim() >= 3:      # probabilities @ values: the kernel's OUTPUT type (attn_out)
            r = rnd(r, Sim.policy.get("attn_out", "f"))
        return r
    return _matmul(rnd(x, a), rnd(y, c))


def p_bmm(x, y):
    a, c = Sim.modes()
    if Sim.region == "bi":                # the two score / two value products of the fusion attention
        a, c = Sim.policy.get("bi_attn", (a, c))
        r = _bmm(rnd(x, a), rnd(y, c))
        if y.shape[-1] == 256 and x.shape[-1] != 256:
            r = rnd(r, Sim.policy.get("attn_out", "f"))
        return r
    return _bmm(rnd(x, a), rnd(y, c))


def p_einsum(eq, x, y):
    a, c = Sim.modes()
    return _einsum(eq, rnd(x, a), rnd(y, c))


def wrap_region(mod, fname, rname):
    orig = getattr(mod, fname)

    def f(*a, **k):
        with region(rname):
            return orig(*a, **k)
    setattr(mod, fname, f)
    return orig


def vit_attention_core_sim(q, k, v, rel_pos_h, rel_pos_w, hw, scale):
    """oracle.ops.vit_attention_core with separately addressable operand roundings: qk (scores), rel (bias), pv."""
    H, W = hw
    with region("vit_attn_qk"):
        attn = p_matmul(q * scale, k.transpose(-2, -1))
    Rh = oo.get_rel_pos(H, H, rel_pos_h)
    Rw = oo.get_rel_pos(W, W, rel_pos_w)
    BH, _, hd = q.shape
    r_q = q.reshape(BH, H, W, hd)
    with region("vit_attn_rel"):
        rel_h = p_einsum("bhwc,hkc->bhwk", r_q, Rh)
        rel_w = p_einsum("bhwc,wkc->bhwk", r_q, Rw)
    attn = (attn.view(BH, H, W, H, W) + rel_h[:, :, :, :, None] + rel_w[:, :, :, None, :]).view(BH, H * W, H * W)
    mode_p = Sim.policy.get("vit_attn_pv", ("f", "f"))
    if mode_p[0] in ("u", "r"):   # unnormalised-P rounding as the flash kernels do: exp(s - max) rounded; row sum of the UNROUNDED p ("u")
        m = attn.max(-1, keepdim=True)[0]           # or of the rounded p ("r": the ones-column row sums of hipie_vit_attn_split)
        p = torch.exp(attn - m)
        ph = rnd(p, "h")
        l = (ph if mode_p[0] == "r" else p).sum(-1, keepdim=True)
        return _matmul(ph, rnd(v, mode_p[1])) / l
    attn = attn.softmax(dim=-1)
    with region("vit_attn_pv"):
        return p_matmul(attn, v)


def run(policy, g, cfg, sd, imgs, ids, mask):
    Sim.policy = policy
    ids, mask = ids[:len(imgs)], mask[:len(imgs)]
    with region("bert"):
        lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", cfg)
    with region("head"):
        out = om.coco_inference(imgs, lang, sd, cfg, task="detection", topk_fg=g["detection_topk_fg"], topk_md=g["detection_topk_md"])
    return {k: rel_err(g.like("detection_" + k, out[k]), g["detection_" + k]) for k in KEYS}


H2 = ("h", "h")
S3 = ("s", "s")          # 3-product split GEMM (lo x lo dropped: below fp32 rounding)
SA = ("s", "h")          # activation split only (2 products)
SW = ("h", "s")
POLICIES = {
    "exact": {},
    # today's "fast": every product on single fp16 operands
    "fast": {"default": H2, "vit_attn_pv": ("u", "h")},
    # today's "parity": fp32 GEMMs, fp16 attention operands
    "parity": {"vit_attn_qk": H2, "vit_attn_rel": H2, "vit_attn_pv": ("u", "h"), "mha": H2},
    "vit_lin_h": {"vit_lin": H2},
    "vit_attn_h": {"vit_attn_qk": H2, "vit_attn_rel": H2, "vit_attn_pv": ("u", "h")},
    "vit_qk_h": {"vit_attn_qk": H2, "vit_attn_rel": H2},
    "vit_pv_h": {"vit_attn_pv": ("u", "h")},
    "vit_all_h": {"vit_lin": H2, "vit_attn_qk": H2, "vit_attn_rel": H2, "vit_attn_pv": ("u", "h")},
    "head_h": {"head": H2, "msda": H2, "mha": H2, "bi": H2},
    "bert_h": {"bert": H2},
    "vit_lin_sa": {"vit_lin": SA},
    "vit_lin_sw": {"vit_lin": SW},
    # candidates for the timed policy
    "split3_all": {"default": S3, "vit_attn_pv": ("u", "s")},
    "split3_attn_h": {"default": S3, "vit_attn_qk": H2, "vit_attn_rel": H2, "vit_attn_pv": ("u", "h")},
    "split3_pv_h": {"default": S3, "vit_attn_pv": ("u", "h")},
    "split3_qk_h": {"default": S3, "vit_attn_qk": H2, "vit_attn_rel": H2, "vit_attn_pv": ("u", "s")},
    "qkv_h": {"vit_qkv": H2}, "proj_h": {"vit_proj": H2}, "fc1_h": {"vit_fc1": H2}, "fc2_h": {"vit_fc2": H2},
    "qkv_sw": {"vit_qkv": SW}, "proj_sw": {"vit_proj": SW}, "fc1_sw": {"vit_fc1": SW}, "fc2_sw": {"vit_fc2": SW},
    "mlp_sw": {"vit_fc1": SW, "vit_fc2": SW},
    "q_split_only": {"vit_attn_qk": ("s", "h"), "vit_attn_rel": ("s", "h")},
    "k_split_only": {"vit_attn_qk": ("h", "s"), "vit_attn_rel": ("h", "s")},
    "pv_ps_vh": {"vit_attn_pv": ("f", "h")},
    "cand_a": {"default": S3, "vit_attn_pv": ("u", "h"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2},
    "cand_bert_attn_h": {"default": S3, "vit_attn_pv": ("u", "h"), "bert_attn": H2},
    "cand_mha_attn_h": {"default": S3, "vit_attn_pv": ("u", "h"), "mha_attn": H2},
    "cand_bi_attn_h": {"default": S3, "vit_attn_pv": ("u", "h"), "bi_attn": H2},
    "cand_einsum_h": {"default": S3, "vit_attn_pv": ("u", "h"), "einsum": H2},
    "cand_einsum_sa": {"default": S3, "vit_attn_pv": ("u", "h"), "einsum": ("s", "h")},
    "cand_b": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2},
    "cand_b_value_h": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2, "lin_out": [("value_proj.", "h")]},
    "cand_b_ffn_h": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2, "lin_out": [("linear1.", "h")]},
    "cand_b_out_h": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2, "attn_out": "h"},
    "cand_b_conv_h": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2, "conv": H2},
    # full-size study (PREC_FIXTURE=e2e_full): which of the remaining single-fp16 spots seeds the decoder's error
    "cand_b_mha_s": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "bi_attn": H2},
    "cand_b_bi_s": {"default": S3, "vit_attn_pv": ("u", "s"), "bert_attn": H2, "mha_attn": H2},
    "cand_b_bert_s": {"default": S3, "vit_attn_pv": ("u", "s"), "mha_attn": H2, "bi_attn": H2},
    "cand_b_pfull": {"default": S3, "vit_attn_pv": ("f", "s"), "bert_attn": H2, "mha_attn": H2, "bi_attn": H2},
    "now": {"default": S3, "vit_attn_pv": ("u", "s"), "bi_attn": H2},            # exact BERT / decoder attention (hipie_attn_f32)
    "now_rl": {"default": S3, "vit_attn_pv": ("r", "s"), "bi_attn": H2},         # ... and row sums of the ROUNDED probabilities (ones column)
    "now_bi_s": {"default": S3, "vit_attn_pv": ("u", "s")},
    "now_p_f": {"default": S3, "vit_attn_pv": ("f", "s"), "bi_attn": H2},
    "only_vit_p_h": {"vit_attn_pv": ("u", "f")},
    "only_mha_h": {"mha_attn": H2},
    "only_bi_h": {"bi_attn": H2},
    "only_bert_h": {"bert_attn": H2},
    "only_split3_lin": {"default": S3},
    # round 4: two-product candidates on top of the shipped policy (PREC_FIXTURE=e2e_full): one ViT linear family with ONE side single fp16
    **{"r4_%s_%s" % (fam, tag): {"default": S3, "vit_attn_pv": ("u", "s"), "vit_" + fam: mode}
       for fam in ("qkv", "proj", "fc1", "fc2") for tag, mode in (("sa", SA), ("sw", SW))},
    "split3_vit_head_h": {"default": H2, "vit_lin": S3, "vit_attn_qk": S3, "vit_attn_rel": S3, "vit_attn_pv": ("u", "s")},
}


def main():
    names = sys.argv[1:] or list(POLICIES)
    fixture = os.environ.get("PREC_FIXTURE", "e2e_deep")
    g = Golden(fixture)
    cfg = g.meta["cfg"]
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist"))
    imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
    ids, mask, _ = _synth.synth_token_ids(2, g.meta["detection"]["n_classes"], g.meta["detection"].get("max_len", 64), seed=74,
                                          pad_to=g.meta["detection"].get("pad_to"))
    # route the oracle's products through the wrappers and name the regions
    F.linear, F.conv2d, F.conv_transpose2d, torch.matmul, torch.bmm, torch.einsum = p_linear, p_conv, p_convt, p_matmul, p_bmm, p_einsum
    torch.Tensor.__matmul__ = lambda a, b: p_matmul(a, b)
    oo.vit_attention_core = vit_attention_core_sim
    real_vit_attention = oo.vit_attention

    def vit_attention(x, sdd, prefix, heads):
        with region("vit_lin"):
            return real_vit_attention(x, sdd, prefix, heads)
    oo.vit_attention = vit_attention
    om.ops.vit_attention = vit_attention
    real_block = om.vit_block

    def vit_block(x, sdd, p, heads, window):
        with region("vit_lin"):
            return real_block(x, sdd, p, heads, window)
    om.vit_block = vit_block
    # output roundings of named linears (storage type of what a GEMM hands on): policy["lin_out"] = [(parameter-prefix suffix, mode)]
    real_lin = om.lin

    def lin_named(x, sdd, p):
        y = real_lin(x, sdd, p)
        for suffix, mode in Sim.policy.get("lin_out", ()):
            if p.endswith(suffix):
                y = rnd(y, mode)
        return y
    om.lin = lin_named
    wrap_region(om, "mha", "mha")
    wrap_region(oo, "bi_attention_block", "bi")
    om.ops.bi_attention_block = oo.bi_attention_block
    wrap_region(oo, "mask_einsum", "einsum")
    om.ops.mask_einsum = oo.mask_einsum
    print("%-22s " % fixture + " ".join("%-9s" % k.replace("pred_", "").replace("maskdino", "md")[:9] for k in KEYS) + "  max")
    for n in names:
        e = run(POLICIES[n], g, cfg, sd, imgs, ids, mask)
        print("%-22s " % n + " ".join("%-9.1e" % e[k] for k in KEYS) + "  %.1e" % max(e.values()), flush=True)


if __name__ == "__main__":
    main()
