"""GPU: the whole product path (HIPIE_IMG.forward_raw -> a22 dict) against the reference-generated golden and the oracle.

The two top-k query selections are discontinuous (a 1-ulp score change swaps queries), so the end-to-end comparison pins
their indices to the reference's (hard part (c)); the free-running selection is checked separately by overlap.
Tolerances (max|a-b| / max|b|, BASELINE.json: "within 1e-3 rel fp16 tolerance"):
  Precision.split3() (the TIMED policy and the registry default: split-fp16 linears and ViT attention, exact small attentions)
                     1e-3 on every a22 output (measured 4e-5 .. 2e-4, incl. the full-size fixture e2e_full);
  Precision.parity() (fp32 library GEMMs, fp16 attention operands)     a SIDE policy, 2e-3 on the end-to-end fixtures: its IoU-head error sits at
                     0.8 - 1.0e-3 and moves with the library's GEMM algorithm from box to box (measured 9.8e-4 on e2e_tiny, 8.4e-4 / 9.5e-4 /
                     1.0e-3 on e2e_r50_512 on three boxes; tools/policy_margins.py prints every measured error beside its bound); 1e-3 on the
                     R50 tiny fixture (measured 3.5e-4) and on the backbone stage; fails at depth: DESIGN.md section 6;
  Precision.fast()   (opt-in: single fp16 operands, fp32 accumulate)   8e-3 on the tiny fixtures (measured 1e-3 .. 6e-3; out of
                     tolerance at the shipped depths -- bench.py prints its numbers as `fast_policy`, never as `value`);
  Precision.bf16()   (bf16 everywhere)                                  8e-2 -- bf16 has 8 mantissa bits, the reference is fp32
                     (measured 7e-3 .. 6e-2)."""
import pytest
import torch

import _synth
from util import Golden, rel_err

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

KEYS = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino",
        "pred_logits_maskdino", "pred_boxes_maskdino"]


def build(precision, fixture="e2e_tiny"):
    from hipie_amd.config import HipieConfig
    from hipie_amd.hipie_img import HIPIE_IMG
    g = Golden(fixture)
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    model = HIPIE_IMG(cfg, precision, device="cuda")
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist"))
    model.load_state_dict(sd, strict=True)
    return g, model.finalize()


def inputs(g, task):
    imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
    ids, mask, pmap = _synth.synth_token_ids(2, g.meta[task]["n_classes"], g.meta[task].get("max_len", 64), seed=74,
                                             pad_to=g.meta[task].get("pad_to"))
    return [{"image": im, "task": task, "input_ids": ids[i], "attention_mask": mask[i],
             "positive_map_label_to_token": pmap} for i, im in enumerate(imgs)]


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_tiny_parity_policy(task):
    from hipie_amd.config import Precision
    g, model = build(Precision.parity())
    model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
    out = model.forward_raw(inputs(g, task))
    errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in KEYS}
    print("parity policy %s: " % task + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 2e-3, (k, errs[k])            # side policy: see the table at the top (9.8e-4 measured, box-dependent)


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_tiny_split_policy(task):
    """the TIMED policy (Precision.split3: split-fp16 GEMMs and attention logits, fp32-class) on the 3-block fixture: 1e-3."""
    from hipie_amd.config import Precision
    g, model = build(Precision.split3())
    model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
    out = model.forward_raw(inputs(g, task))
    errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in KEYS}
    print("split policy %s: " % task + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 1e-3, (k, errs[k])


def test_e2e_deep_split_policy():
    """the TIMED policy at the SHIPPED depths (32 ViT blocks with the real window pattern, 6 + 6 encoder / decoder layers, 6 + 9
    MaskDINO layers, FFN 2048, 900 + 10 / 300 queries, 12-layer BERT; tests/golden/e2e_deep.npz from the reference's own
    coco_inference): every a22 output within 1e-3.  (The parity policy -- fp32 GEMMs but single-fp16 attention logits -- measures
    2.5e-3 here, the fp16 'fast' policy 1.2e-2: tools/deep_err.py.)"""
    from hipie_amd.config import Precision
    g, model = build(Precision.split3(), "e2e_deep")
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    out = model.forward_raw(inputs(g, "detection"))
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("split policy, full depth: " + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 1e-3, (k, errs[k])


@pytest.mark.parametrize("policy,fixture", [("split3", "e2e_full"), ("split3", "e2e_full_refinit"), ("mixed", "e2e_full_refinit")])
def test_e2e_full_size_split_policy(policy, fixture):
    """(e2e_full_refinit: the same configuration with weights drawn from the reference's OWN initialisation distribution,
    tests/golden/refinit_stats.json; `mixed` = split linears + single-fp16 ViT attention core, in tolerance on that distribution only.)
    BASELINE.json's headline configuration itself -- the full ViT-H (1280 wide, 16 heads, 32 blocks, 64 x 64 token grid) with the shipped
    head sizes on a 1024 x 1024 image -- against tests/golden/e2e_full.npz, the reference's own coco_inference on the CPU with the same
    synthetic weights (outputs above 65536 elements compared on the fixture's strided subsample): every a22 output within 1e-3 in
    the TIMED policy.  This is the arithmetic the throughput is quoted on, at the size it is quoted on."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", fixture + ".npz")):
        pytest.skip("tests/golden/%s.npz not generated" % fixture)
    from hipie_amd.config import Precision
    g, model = build(getattr(Precision, policy)(), fixture)
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    out = model.forward_raw(inputs(g, "detection")[:len(g.meta["sizes"])])
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("%s policy, FULL SIZE (ViT-H, 1024^2) on %s: " % (policy, fixture) + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 1e-3, (k, errs[k])


@pytest.mark.parametrize("policy,tol", [("split3", 1e-3), ("parity", 1e-3), ("fast", 1e-2)])       # fast: measured 1e-3 .. 8e-3 (pred_masks)
def test_e2e_long_prompt(policy, tol):
    """BASELINE configs[3]-style prompt inside the FULL path: 815 tokens go through BertEncoder's > 512 chunker
    (bert_model.py:61-135), are padded to 896 (PAD_MAX), and the fusion / class-logit kernels run over L = 896 with a
    half-masked tail; against the reference's own coco_inference on the same inputs."""
    from hipie_amd.config import Precision
    g, model = build(getattr(Precision, policy)(), "e2e_long_tiny")
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    batch = inputs(g, "detection")
    assert batch[0]["input_ids"].shape[0] == 896 and int(batch[0]["attention_mask"].sum()) > 512
    out = model.forward_raw(batch)
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("long prompt, %s policy: " % policy + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < tol, (k, errs[k])


def test_windowed_qkv_buffers_persistent_and_shared_modes_agree():
    """the window-layout qkv buffers of the windowed ViT blocks (vit.QKV_BUFFERS): per block while the process-wide budget allows, one shared
    scratch buffer refilled by hipie_fill_rows beyond it (large per-GPU batches) -- bit-identical outputs, bounded memory, and a second
    geometry does not leave the first one's buffers behind."""
    from hipie_amd.config import Precision
    from hipie_amd.modeling import vit
    g, model = build(Precision.split3())
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    batch = inputs(g, "detection")
    saved = vit.QKV_BUFFERS.budget
    try:
        vit.QKV_BUFFERS.budget = 8 << 30
        model.forward_raw(batch)                                               # (the first forward builds the per-geometry constants)
        a = model.forward_raw(batch)
        blocks = [m for m in model.modules() if isinstance(m, vit.Attention) and m.__dict__.get("_qkv_state") is not None]
        assert len(blocks) == 2 and vit.QKV_BUFFERS._live_bytes() > 0          # the tiny ViT has two windowed blocks
        a2 = model.forward_raw(batch)                                          # control: run-to-run spread of the persistent mode itself
        one = [dict(batch[0])]                                                  # a second geometry (one image): the old buffers are replaced
        model.pin_topk(None, None)                                              # (the pinned indices belong to the two-image canvas)
        model.forward_raw(one)
        torch.cuda.synchronize()
        n_geo = len(set(e[1] for e in vit.QKV_BUFFERS.entries.values() if e[0]() is not None and e[0]() in blocks))
        assert n_geo == 1
        model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
        vit.QKV_BUFFERS.budget = 0                                              # nothing fits: the shared scratch buffer
        for m in blocks:
            m._qkv_state = None
        vit.QKV_BUFFERS.entries.clear()
        b = model.forward_raw(batch)
        assert all(m.__dict__.get("_qkv_state") is None for m in blocks) and len(vit.QKV_BUFFERS.shared) == 1
        for k in KEYS:
            scale = float(a[k].abs().max())
            ctrl = float((a2[k] - a[k]).abs().max())
            diff = float((b[k] - a[k]).abs().max())
            print("%s: shared vs persistent %.2e, persistent run-to-run %.2e (scale %.2e)" % (k, diff, ctrl, scale))
            assert diff <= max(2 * ctrl, 1e-6 * scale), k
    finally:
        vit.QKV_BUFFERS.budget = saved
        vit.QKV_BUFFERS.shared.clear()


def test_windowed_qkv_buffers_follow_the_geometry_not_the_size():
    """token grids 50 x 76 and 50 x 72 both pad to 56 x 84 = 4704 window rows at ws 14, with their padding rows in
    DIFFERENT places.  A persistent qkv buffer keyed on (rows, width) alone kept the first image's qkv in rows that are padding for the
    second (they entered the window softmax as keys).  Alternating the two geometries on one model must give, for each, exactly what a
    model that has only ever seen that geometry gives."""
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.modeling import vit
    from hipie_amd.modeling.vit import D2ViT
    g = Golden("vit_backbone")
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    sd = _synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=31)

    def fresh():
        m = D2ViT(cfg, Precision.split3())
        m.load_state_dict(sd)
        return m.cuda().eval().cast_weights()

    xa = _synth.synth_tensor("vit_in_a", (1, 3, 800, 1216), seed=33).cuda()       # 50 x 76 tokens
    xb = _synth.synth_tensor("vit_in_b", (1, 3, 800, 1152), seed=34).cuda()       # 50 x 72 tokens: same 4704 window rows
    saved = vit.QKV_BUFFERS.budget
    try:
        vit.QKV_BUFFERS.budget = 8 << 30
        want_a, want_b = fresh()(xa), fresh()(xb)
        m = fresh()
        got = [m(xa), m(xb), m(xa), m(xb)]
        torch.cuda.synchronize()
        for k in ("res3", "res4", "res5"):
            for i, want in enumerate((want_a, want_b, want_a, want_b)):
                assert torch.equal(got[i][k], want[k]), (k, i, float((got[i][k] - want[k]).abs().max()))
    finally:
        vit.QKV_BUFFERS.budget = saved


def test_e2e_pad_max_4096_is_trimmed():
    """The shipped eval setting (MODEL.LANGUAGE_BACKBONE.PAD_MAX, MAX_QUERY_LEN 4096: configs/eval/image_joint_vit_huge_32g_pan_maskdino_ade_test.yaml:10-11,
    hipie_img.py:904-909) in the TIMED policy against the reference's own coco_inference on the same 4096-column inputs
    (tests/golden/e2e_padmax_tiny.npz): every a22 output -- the 4096-column class logits included -- within 1e-3, AND the padding is free:
    the text encoder hands the fusion the real tokens plus one padding row (their hidden states are zero in the reference,
    bert_model.py:118-127, so all padding rows / columns are alike), the image -> text softmax runs over the attended keys only
    (masked keys have probability exactly 0, fuse_helper.py:96-109), and nothing of size Nv x 4096 is allocated."""
    from hipie_amd.config import Precision
    g, model = build(Precision.split3(), "e2e_padmax_tiny")
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    batch = inputs(g, "detection")
    assert batch[0]["input_ids"].shape[0] == 4096 and int(batch[0]["attention_mask"].sum()) < 64
    short = [dict(b, input_ids=b["input_ids"][:64], attention_mask=b["attention_mask"][:64]) for b in batch]
    model.forward_raw(short)                       # same weights, same images, the caption unpadded: the memory yard-stick
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    model.forward_raw(short)
    torch.cuda.synchronize()
    peak_short = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    out = model.forward_raw(batch)
    torch.cuda.synchronize()
    peak_pad = torch.cuda.max_memory_allocated()
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("PAD_MAX 4096, split policy: " + " ".join("%s=%.1e" % kv for kv in errs.items()))
    assert out["pred_logits"].shape[-1] == 4096 and out["pred_logits_maskdino"].shape[-1] == 4096
    for k in KEYS:
        assert errs[k] < 1e-3, (k, errs[k])
    B = len(batch)
    Nv = sum((256 // s) ** 2 for s in (8, 16, 32, 64))
    one_score_tensor = B * 8 * Nv * 4096 * 4             # S (fp32) = P (HL8) bytes of the untrimmed image -> text direction
    print("peak memory: unpadded %.1f MB, PAD_MAX %.1f MB (an Nv x 4096 score tensor would be %.1f MB)"
          % (peak_short / 2 ** 20, peak_pad / 2 ** 20, one_score_tensor / 2 ** 20))
    assert peak_pad - peak_short < one_score_tensor // 8


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_full_size_bench_inputs(task):
    """the workload bench.py TIMES, literally (BASELINE configs[2]): full ViT-H, shipped head sizes, image 0 of bench.synth_batch at
    1024 x 1024, the 80-class caption (L = 194) -- and the separate grounding call (one referring expression) -- in the timed policy
    against tests/golden/e2e_full_c80.npz = the reference's own coco_inference on those inputs."""
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", "e2e_full_c80.npz")):
        pytest.skip("tests/golden/e2e_full_c80.npz not generated")
    import bench
    from hipie_amd.config import Precision
    g, model = build(Precision.split3(), "e2e_full_c80")
    bi = g.meta["bench_inputs"]
    batch = bench.synth_batch(None, 1, bi["size"], bi["n_classes"], bi["L"], "cpu", seed=bi["seed"], task=task)
    assert batch[0]["input_ids"].shape[0] == g.meta[task]["L"]
    model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
    out = model.forward_raw(batch)
    errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in KEYS}
    print("split3, FULL SIZE, bench inputs (%s, L = %d): " % (task, g.meta[task]["L"]) + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 1e-3, (k, errs[k])


@pytest.mark.parametrize("policy", ["split3", "parity"])          # split3 = the registry default / the timed policy
def test_e2e_tiny_free_topk_overlap(policy):
    """un-pinned selection: the product's own two-stage top-k (hipie_topk) picks >= 90 % of the reference's queries"""
    from hipie_amd.config import Precision
    g, model = build(getattr(Precision, policy)())
    model.forward_raw(inputs(g, "detection"))
    fg, md = model.last_topk()
    for got, want in ((fg.cpu(), g["detection_topk_fg"]), (md.cpu(), g["detection_topk_md"])):
        for b in range(got.shape[0]):
            inter = len(set(got[b].tolist()) & set(want[b].tolist()))
            assert inter >= 0.9 * want.shape[1]


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_tiny_fast_policy(task):
    """the opt-in `fast` policy (single fp16 operands, fp32 accumulation and residual stream; NOT the timed one): every a22 output within 8e-3
    (a 16-bit operand pipeline of this depth does not reach the parity policy's 1e-3: tools/prec_matrix.py shows every stage
    contributing 1-3e-3; bf16 is at 2-3e-2)."""
    from hipie_amd.config import Precision
    g, model = build(Precision.fast())
    model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
    out = model.forward_raw(inputs(g, task))
    errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in KEYS}
    print("fast policy %s: " % task + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 8e-3, (k, errs[k])


def test_e2e_tiny_bf16_policy():
    from hipie_amd.config import Precision
    g, model = build(Precision.bf16())
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    out = model.forward_raw(inputs(g, "detection"))
    errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
    print("bf16 policy: " + " ".join("%s=%.1e" % kv for kv in errs.items()))
    for k in KEYS:
        assert errs[k] < 8e-2, (k, errs[k])


@pytest.mark.parametrize("task", ["detection", "grounding"])
def test_e2e_r50_tiny(task):
    """the R50 configs: MIOpen ResNet-50 + the same HIP heads, against the reference's own coco_inference.  `grounding` = BASELINE
    configs[0] (one image batch + ONE text prompt), `detection` = configs[1] (class prompts).  Split policy (the default of the
    drop-in) and the fp32 parity policy inside 1e-3; the fast / bf16 policies at their own bounds."""
    from hipie_amd.config import Precision
    for prec, tol in ((Precision.split3(), 1e-3), (Precision.parity(), 1e-3), (Precision.fast(), 8e-3), (Precision.bf16(), 8e-2)):
        g, model = build(prec, "e2e_r50_tiny")
        model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
        out = model.forward_raw(inputs(g, task))
        for k in KEYS:
            assert rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) < tol, (k, str(prec))


@pytest.mark.parametrize("task", ["grounding", "detection"])
def test_e2e_r50_512_literal_config0(task):
    """BASELINE configs[0] LITERALLY (`grounding`): the reference's R50 with the SHIPPED head sizes on ONE 512 x 512 image with ONE
    referring expression, against the reference's own CPU coco_inference (tests/golden/e2e_r50_512.npz, about half a minute of
    MODEL.DEVICE = cpu); `detection` = the same model with a class prompt.  The timed (split3) policy -- the drop-in's default -- is held
    at the north star's 1e-3 (measured 2e-5).  The parity policy keeps fp16 ATTENTION operands (config.Precision.parity) and its fp32
    linears run on the library's GEMMs, whose algorithm (and so the rounding) differs from box to box: its IoU-head error was measured
    at 8.4e-4, 9.5e-4 and 1.0e-3 on three boxes of the pool, so its bound here is 2e-3 -- it is a side policy, not the shipped path."""
    from hipie_amd.config import Precision
    for prec, tol in ((Precision.split3(), 1e-3), (Precision.parity(), 2e-3)):
        g, model = build(prec, "e2e_r50_512")
        assert tuple(g.meta["sizes"][0]) == (512, 512) and len(g.meta["sizes"]) == 1
        model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
        out = model.forward_raw(inputs(g, task))
        errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in KEYS}
        print("configs[0] literal (R50, 512^2, %s), %s policy: " % (task, prec.name) + " ".join("%s=%.1e" % kv for kv in errs.items()))
        for k in KEYS:
            assert errs[k] < tol, (k, errs[k], str(prec))


def test_full_size_r50_bs4():
    """BASELINE configs[1] at FULL SIZE inside the suite: R50, 1024 x 1024, batch 4, the 80-class caption (L = 194), shipped head sizes, timed
    policy.  No reference run exists at this size (the fixture above pins the same model at 512^2), so the checks are the size-independent
    ones: finite outputs of the a22 shapes, and batch exchange -- image i's rows do not depend on its neighbours (pinned top-k)."""
    import bench
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    g = Golden("e2e_r50_512")
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    model = HIPIE_IMG(cfg, Precision.split3(), device="cuda")
    sd = _synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist"))
    model.load_state_dict(sd, strict=True)
    model.finalize()
    batch = bench.synth_batch(None, 4, 1024, 80, 194, "cuda", seed=0, task="detection")
    out = model.forward_raw(batch)
    fg, md = model.last_topk()
    assert out["pred_logits"].shape[:2] == (4, 910) and out["pred_masks"].shape[0] == 4 and out["pred_masks_maskdino"].shape[:2] == (4, 300)
    for k in KEYS:
        assert torch.isfinite(out[k].float()).all(), k
    perm = [2, 0, 3, 1]
    model.pin_topk(fg[perm], md[perm])
    out2 = model.forward_raw([batch[i] for i in perm])
    for k in KEYS:
        a, b = out[k].float()[perm], out2[k].float()
        assert rel_err(b.cpu(), a.cpu()) < 1e-4, (k, rel_err(b.cpu(), a.cpu()))


def test_stage_vit_backbone():
    """D2ViT alone (3 blocks: window, window, global; interpolated abs-pos and rel-pos tables) vs the reference golden."""
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.modeling.vit import D2ViT
    g = Golden("vit_backbone")
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    for prec, tol in ((Precision.split3(), 2e-4), (Precision.parity(), 1e-3), (Precision.fast(), 3e-3), (Precision.bf16(), 3e-2)):
        m = D2ViT(cfg, prec)
        m.load_state_dict(_synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=31))
        m = m.cuda().eval().cast_weights()
        x = _synth.synth_tensor("vit_in", g.meta["x_shape"], seed=32).cuda()
        out = m(x)
        for k in ("res3", "res4", "res5"):
            assert rel_err(g.like(k, out[k].float().cpu()), g[k]) < tol, (k, prec)


def test_stage_bert_short_and_chunked():
    """product BertEncoder (<= 512 tokens and the > 512 chunking path, bert_model.py:32-153) on the GPU vs the golden made by
    the reference's BertEncoder + transformers.BertModel."""
    from hipie_amd.config import HipieConfig
    from hipie_amd.modeling.text import BertEncoder
    g = Golden("bert")
    cfg = HipieConfig.from_dict(g.meta["cfg"])
    m = BertEncoder(cfg)
    m.load_state_dict(_synth.synth_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, seed=51))
    m = m.cuda().eval()
    for tag in ("short", "long"):
        out = m({"input_ids": g[tag + "_ids"].cuda(), "attention_mask": g[tag + "_mask"].cuda()}, sep=1012)["hidden"]
        assert rel_err(g.like(tag + "_hidden", out.float().cpu()), g[tag + "_hidden"]) < 2e-4


@pytest.mark.parametrize("policy", ["split3", "parity"])
def test_e2e_batch_items_are_independent(policy):
    """size-independent property: with equal image sizes the a22 rows of an image do not depend on its batch neighbours
    (exercises every batch / head / XCD index map of the kernels); pinned top-k so the comparison is continuous."""
    from hipie_amd.config import Precision
    g, model = build(getattr(Precision, policy)())
    imgs = _synth.synth_images([(192, 256), (192, 256), (192, 256)], seed=99)
    ids, mask, pmap = _synth.synth_token_ids(3, 9, 64, seed=74)

    def batch(idx):
        return [{"image": imgs[i], "task": "detection", "input_ids": ids[0], "attention_mask": mask[0],
                 "positive_map_label_to_token": pmap} for i in idx]
    full = model.forward_raw(batch([0, 1, 2]))
    fg, md = model.last_topk()
    for i in (0, 2):
        model.pin_topk(fg[i:i + 1].cpu(), md[i:i + 1].cpu())
        one = model.forward_raw(batch([i]))
        for k in KEYS:
            assert rel_err(one[k].float().cpu(), full[k][i:i + 1].float().cpu()) < 1e-3, (k, i)
    model.pin_topk(None, None)


@pytest.mark.parametrize("policy,n_classes,L,sizes", [
    ("split3", 150, 815, [(1024, 1024)] * 2),                      # configs[3]: a rank's shard of the bs-64 job -- 1024^2, ADE-150 prompt
    ("split3", 1203, 4096, [(1000, 1333), (1333, 1000)]),          # configs[4]: 1333-pixel long edge inside a 1344^2 canvas (real pad masks)
    ("fast", 1203, 4096, [(1344, 1344)] * 2)])
def test_full_size_open_vocabulary_configs(policy, n_classes, L, sizes):
    """BASELINE.json configs[3] / [4] at full size on ViT-H: the ADE-150 caption of ~815 tokens on 1024^2 images (what one rank of
    the 8-GPU bs-64 job runs) and the LVIS-1203 caption cut at MAX_QUERY_LEN 4096 (chunked BERT) on 1333-pixel long-edge images --
    padded by the product into a 1344^2 canvas, i.e. an 84x84 global-attention grid (the NB = 3 kernels) WITH pad masks through
    position embedding, encoder and proposals.  No golden at this size (the reference cannot run here in seconds): size-independent
    properties instead -- every a22 output finite, and exchanging the two images of the batch exchanges their rows."""
    import bench
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, getattr(Precision, policy)(), device="cuda")
    bench.randomize_degenerate_inits(model)
    model.finalize()
    b = bench.synth_batch(cfg, 2, max(sizes[0]), n_classes, L, torch.device("cuda"))
    for item, (h, w) in zip(b, sizes):
        item["image"] = item["image"][:, :h, :w].contiguous()
    n_named = len(b[0]["positive_map_label_to_token"])
    assert n_named == n_classes and int(b[0]["attention_mask"].sum()) > 0.85 * L
    out = model.forward_raw(b)
    fg, md = model.last_topk()
    canvas = 32 * ((max(max(s) for s in sizes) + 31) // 32)
    assert out["pred_masks"].shape[-2:] == (canvas // 4, canvas // 4)
    for k in KEYS:
        assert torch.isfinite(out[k].float()).all(), k
    model.pin_topk(fg.flip(0).cpu(), md.flip(0).cpu())
    swapped = model.forward_raw(b[::-1])
    model.pin_topk(None, None)
    for k in KEYS:
        # fast: the library GEMMs' reduction order depends on the row position (~1e-3); split3: hipie_gemm's order is fixed, what is
        # left are MIOpen's convolutions
        assert rel_err(swapped[k].float().flip(0).cpu(), out[k].float().cpu()) < (1e-2 if policy == "fast" else 1e-3), k
    # the instance post-processing sees the whole vocabulary: one score per (query, class)
    from hipie_amd.postprocess import inference
    res = inference(model, out, b, with_masks=False, with_sem_pan=False)
    assert int(res[0]["instances"].pred_classes.max()) < n_classes


def test_full_size_wide_image_split_policy():
    """the shipped eval yamls resize to MIN_SIZE_TEST 1024 / MAX_SIZE_TEST 2048: a 2:1 image becomes 1024 x 2048 = a 64 x 128 token grid,
    wider than the 96 key slots of a global-attention tile.  The registry-default policy (split3) walks such grids column by column
    (hipie_vit_attn_split, transposed form; reference-generated golden `global128` in tests/test_gpu_kernels.py).  Full-size ViT-H,
    size-independent properties: shapes, finite outputs, batch-exchange equivariance with a real pad mask in the second image."""
    import bench
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, Precision.split3(), device="cuda")
    bench.randomize_degenerate_inits(model)
    model.finalize()
    sizes = [(1024, 2048), (1000, 1920)]
    b = bench.synth_batch(cfg, 2, 2048, 80, 194, torch.device("cuda"))
    for item, (h, w) in zip(b, sizes):
        item["image"] = item["image"][:, :h, :w].contiguous()
    out = model.forward_raw(b)
    fg, md = model.last_topk()
    assert out["pred_masks"].shape[-2:] == (256, 512) and out["pred_masks_maskdino"].shape[-2:] == (256, 512)
    for k in KEYS:
        assert torch.isfinite(out[k].float()).all(), k
    model.pin_topk(fg.flip(0).cpu(), md.flip(0).cpu())
    swapped = model.forward_raw(b[::-1])
    model.pin_topk(None, None)
    for k in KEYS:
        assert rel_err(swapped[k].float().flip(0).cpu(), out[k].float().cpu()) < 1e-3, k


def test_split_policy_large_activations_stay_finite():
    """scaled weights: the linears of the backbone and the heads see activations 30x larger than the synthetic checkpoint's (real
    checkpoints have outlier channels).  The split policy's operands are fp16 PAIRS: values beyond the fp16 range saturate (never inf /
    NaN), everything else keeps fp32-class accuracy -- every a22 output stays finite."""
    from hipie_amd.config import Precision
    g, model = build(Precision.split3())
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("mlp.fc1.weight") or n.endswith("linear1.weight") or "input_proj" in n and n.endswith("0.weight"):
                p.mul_(30.0)
    model._invalidate()
    out = model.forward_raw(inputs(g, "detection"))
    for k in KEYS:
        assert torch.isfinite(out[k].float()).all(), k


@pytest.mark.parametrize("policy,e2e,stages", [("split3", "e2e_tiny", "stages_tiny"), ("parity", "e2e_tiny", "stages_tiny"),
                                               ("split3", "e2e_full", "stages_full")])
def test_stages_on_the_gpu(policy, e2e, stages):
    """the PRODUCT path stage by stage against tests/golden/stages_tiny.npz (intermediate tensors of the reference's own modules
    inside coco_inference): backbone features + sine position + level masks (rows a3, a7), the encoder memory and the fused
    language stream (a9, a11), decoder states and references (a13; the two-stage selection a12 pinned), the CondInst mask-head
    convolutions (a18), MaskDINO's encoder memory, multi-scale features and mask features (a8, a20).  Forward hooks on the modules;
    the a15 / a16 / a17 heads are the last step from `dec_hs` to the a22 outputs checked by the e2e tests."""
    from hipie_amd.config import Precision
    import os
    if not os.path.exists(os.path.join(os.path.dirname(__file__), "golden", stages + ".npz")):
        pytest.skip("tests/golden/%s.npz not generated" % stages)
    st = Golden(stages)           # stages_full: the same tensors at the headline configuration (full ViT-H, one 1024 x 1024 image)
    g, model = build(getattr(Precision, policy)(), e2e)
    model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
    d = model.detr
    caps = {}
    bad = {}

    def cap(name):
        def f(mod, inp, out):
            caps[name] = out
        return f
    hooks = [d.detr.backbone.register_forward_hook(cap("backbone")),
             d.detr.transformer.encoder.register_forward_hook(cap("encoder")),
             d.detr.transformer.decoder.register_forward_hook(cap("decoder")),
             d.mask_head.register_forward_hook(cap("mask_head")),
             d.mask_dino.pixel_decoder.transformer.register_forward_hook(cap("md_enc"))]
    pix = d.mask_dino.pixel_decoder
    orig_ff = pix.forward_features

    def ff(features, masks=None):
        r = orig_ff(features, masks)
        caps["md_pix"] = r
        return r
    pix.forward_features = ff
    try:
        model.forward_raw(inputs(g, "detection")[:len(g.meta["sizes"])])
    finally:
        pix.forward_features = orig_ff
        for h in hooks:
            h.remove()
    tol = 1e-3

    def chk(key, t, valid=None):
        """max|a-b| / max|b| against the fixture; `valid` (bool, broadcastable to t): compare those entries only."""
        a, b = st.like(key, t.float().cpu().contiguous()), st[key]
        if valid is None:
            e = rel_err(a, b)
        else:
            v = st.like(key, valid.expand_as(t).contiguous().cpu())
            e = float(((a - b).abs() * v).max() / (b.abs() * v).max())
        if not e < tol:
            bad[key] = e
        return e
    feats, pos = caps["backbone"]
    errs = {}
    for i in range(3):
        errs["feat%d" % i] = chk("feat%d" % i, feats[i].tensors)
        assert torch.equal(st.like("mask%d" % i, feats[i].mask.cpu()), st["mask%d" % i])
        # the sine embedding is compared on the VALID pixels: in a fully padded row the x coordinate is (0 - 0.5) / 1e-6 and the
        # embedding is sin / cos of ~3e6 rad -- a value nothing reads (the padding mask removes it), on which the host's and the
        # device's range reduction legitimately differ
        valid = (~feats[i].mask.cpu())[:, None].expand_as(pos[i]).contiguous()
        a = st.like("pos%d" % i, pos[i].float().cpu().contiguous())
        v = st.like("pos%d" % i, valid)
        e = float(((a - st["pos%d" % i]).abs() * v).max())
        if not e < 1e-4:
            bad["pos%d" % i] = e
        errs["pos%d" % i] = e
    # PADDED tokens carry the sine embedding of a degenerate coordinate (see above): the reference's values there depend on its
    # device's sin / cos of ~3e6 rad, and nothing reads them except the 3x3 convolutions of the mask head, which smear them a few
    # pixels into the image.  The encoder memory is compared on the valid tokens, the mask-head map 9 pixels away from the padding
    # (receptive field of lay3@s32 .. lay2@s8 in stride-8 pixels).
    m0 = feats[0].mask
    lvl_masks = [f.mask for f in feats] + [torch.nn.functional.interpolate(m0[None].float(), size=(m0.shape[1] // 8, m0.shape[2] // 8)).bool()[0]]
    tok_valid = ~torch.cat([m.flatten(1) for m in lvl_masks], 1)
    errs["memory"] = chk("memory", caps["encoder"]["visual"], tok_valid[..., None])
    errs["vl0_lang"] = chk("vl0_lang", caps["encoder"]["lang"]["hidden"])
    hs, refs = caps["decoder"][0], caps["decoder"][1]
    errs["dec_hs"] = chk("dec_hs", hs)
    errs["dec_refs"] = chk("dec_refs", refs)
    far = torch.nn.functional.max_pool2d(m0[:, None].float(), 19, 1, 9)[:, 0] == 0            # no padded pixel within 9 pixels
    errs["mask_head"] = chk("mask_head_out", caps["mask_head"], far[:, None])
    errs["md_enc_memory"] = chk("md_enc_memory", caps["md_enc"][0])
    mf, _, ms = caps["md_pix"]
    if not torch.is_tensor(mf):                  # 16-bit policies keep the mask features in front of their last 1x1 convolution
        mf = torch.nn.functional.conv2d(mf.pre.float(), mf.weight.float()[:, :, None, None], mf.bias.float())
    errs["md_mask_features"] = chk("md_mask_features", mf)
    for i in range(4):
        errs["md_ms%d" % i] = chk("md_ms%d" % i, ms[i])
    print("stages %s (%s): " % (stages, policy) + " ".join("%s=%.1e" % kv for kv in errs.items()))
    assert not bad, bad


def test_two_stream_step_repeats_bit_for_bit_at_full_size():
    """The timed step (ViT-H, 1024^2, bs 8, split3) runs its two head branches on two streams (modeling/ddetrs_dn.py); until round 6 that made
    the outputs move by ~1e-3 from run to run (the packed-fp32 erratum behind hipie_msda_fused).  Every hand-written kernel of
    the path is deterministic (fixed accumulation order, no atomics), so with the top-k selections pinned the step must now repeat BIT FOR BIT
    up to the run-to-run spread the library convolutions have on ONE stream as well (measured: fp32 rounding level) -- five repeats each way.  A kernel
    of ours OR of a library that mis-executes beside the other branch's GEMMs (25 % of the launches did) shows up here."""
    import bench
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, Precision.split3(), device="cuda")
    bench.randomize_degenerate_inits(model)
    model.finalize()
    assert model.detr.overlap_branches
    b = bench.synth_batch(cfg, 8, 1024, 80, 194, torch.device("cuda"))
    first = model.forward_raw(b)
    fg, md = model.last_topk()
    model.pin_topk(fg.cpu(), md.cpu())
    dev_by_mode = {}
    for overlap in (True, False):
        model.detr.overlap_branches = overlap
        ref = {k: v.float().clone() for k, v in model.forward_raw(b).items() if torch.is_tensor(v)}
        worst = {k: 0.0 for k in KEYS}
        for _ in range(5):
            out = model.forward_raw(b)
            torch.cuda.synchronize()
            for k in KEYS:
                worst[k] = max(worst[k], float((out[k].float() - ref[k]).abs().max() / ref[k].abs().max()))
        dev_by_mode[overlap] = worst
        print("%s, 5 repeats, max relative deviation per output: " % ("two streams" if overlap else "one stream ") + " ".join("%s=%.1e" % kv for kv in worst.items()))
    model.detr.overlap_branches = True
    # what is left is the run-to-run spread of the library convolutions (MIOpen picks among equivalent solutions / accumulation orders), the
    # same with one stream as with two; a mis-executed kernel moves an output by 1e-3 (measured before the fix), two orders above the bound
    for k in KEYS:
        assert dev_by_mode[True][k] < 5e-5, (k, dev_by_mode[True][k])
        assert dev_by_mode[True][k] <= 10 * dev_by_mode[False][k] + 1e-5, (k, dev_by_mode)


def test_two_stream_forward_replays_in_a_hip_graph():
    """the forward with its side stream (the MaskDINO head beside the deformable transformer) captured ONCE into a hipGraph -- the fork / join
    of the branch become graph dependencies -- and replayed: the replays equal the eager forward (bench.py --graph does this at full size)."""
    from hipie_amd.config import Precision
    g, model = build(Precision.split3())
    model.pin_topk(g["detection_topk_fg"].cuda(), g["detection_topk_md"].cuda())
    batch = inputs(g, "detection")
    for b in batch:                                         # images and token ids on the device: nothing crosses the PCIe inside the capture
        b["image"], b["input_ids"], b["attention_mask"] = b["image"].cuda(), b["input_ids"].cuda(), b["attention_mask"].cuda()
    eager = {k: v.float().clone() for k, v in model.forward_raw(batch).items() if torch.is_tensor(v)}
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model.forward_raw(batch)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = model.forward_raw(batch)
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    for k in KEYS:
        assert rel_err(out[k].float().cpu(), eager[k].cpu()) < 2e-5, (k, rel_err(out[k].float().cpu(), eager[k].cpu()))
