#!/usr/bin/env python3
"""bench.py -- images/sec of the HIPIE single-image inference hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path (HIPIE_IMG.forward_raw: preprocess -> BERT -> ViT -> fused deformable transformer ->
MaskDINO -> a22 output dict, then the compact per-image top-100 predictions and -- for N > 1 -- their RCCL all-gather)
over one batch of synthetic images that is already resident in HBM.  Workload (BASELINE.json configs[2], the one the
metric is quoted on): ViT-H backbone, 1024x1024, batch 8 per GPU, COCO-80 class prompt (L = 194 tokens), task "detection".
Data parallel (weak scaling): ONE global batch of 8 * N images (image i is seeded by i) is cut into contiguous index ranges with
parallel.shard_range -- configs[3] literally at N = 8: 64 images, 8 per rank -- and every rank's compact predictions are
all-gathered (RCCL over xGMI); the JSON line carries the number of ranks the collective really saw and the gathered block's shape.

The timed arithmetic is Precision.split3 ("split"): every linear and the ViT attention logits on split-fp16 operands (three MFMA
products, fp32 accumulation = fp32-class), the policy whose a22 outputs stay within BASELINE.json's 1e-3 of the reference at the
shipped depths (`parity_err`, measured by this script on tests/golden/e2e_deep.npz and e2e_tiny.npz).  `fast_policy` is the
out-of-tolerance fp16 throughput mode, timed beside it for reference only.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant hand-written kernel (gemm_kernel<320, split>: the four ViT
linears, 128 launches per step), timed live with HIP events on the launch stream; `roofline_attention` the same for the global
ViT attention kernel; `cpu_baseline` times the oracle (CPU restatement of the reference) on this host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def synth_image(index, size, device, seed=0):
    """image `index` of the global synthetic batch: uint8-valued, seeded by its GLOBAL index (every rank can build any image)."""
    gi = torch.Generator().manual_seed(1000003 * seed + 7919 * index + 13)
    return torch.randint(0, 256, (3, size, size), generator=gi).float().to(device)


def synth_batch(cfg, batch, size, n_classes, L, device, seed=0, task="detection", indices=None, ids_device="cpu"):
    """`batch` inputs; images are those of the global indices `indices` (default 0 .. batch-1), the prompt is shared.  Images are
    placed on `device`; token ids / masks stay on the HOST, where the reference's tokenizer produces them (hipie_img.py:904-909) -- the
    text encoder copies them over inside the step (ids_device: put them elsewhere, e.g. on the GPU for a hipGraph capture)."""
    g = torch.Generator().manual_seed(seed)
    idev = ids_device
    indices = list(range(batch)) if indices is None else list(indices)
    if task == "grounding":                     # one referring expression of ~10 tokens (BASELINE configs[2], second call)
        n = 10
        ids = torch.tensor([101] + torch.randint(1996, 29000, (n,), generator=g).tolist() + [102])
        mask = torch.ones_like(ids)
        return [{"image": synth_image(i, size, device, seed), "task": "grounding",
                 "input_ids": ids.to(idev), "attention_mask": mask.to(idev)} for i in indices]
    ids = torch.zeros(L, dtype=torch.long)
    mask = torch.zeros(L, dtype=torch.long)
    row, pmap = [101], {}
    # 80 classes: the COCO caption (1-3 tokens a name, 194 tokens; a longer L is PAD_MAX padding).  Other vocabularies
    # (BASELINE configs[3]/[4]: ADE-150 at L~815, LVIS-1203 cut at MAX_QUERY_LEN 4096) get names sized so the caption fills L.
    cap = min(L, 194) if n_classes == 80 else L
    kmean = 2 if n_classes == 80 else max(1, round((L - 2) / n_classes) - 1)
    for c in range(n_classes):
        k = int(torch.randint(max(1, kmean - 1), kmean + 2, (1,), generator=g))
        if len(row) + k + 2 > cap:
            break
        pmap[c + 1] = list(range(len(row), len(row) + k))
        row += torch.randint(1996, 29000, (k,), generator=g).tolist() + [1012]
    row.append(102)
    ids[:len(row)] = torch.tensor(row)
    mask[:len(row)] = 1
    out = []
    for i in indices:
        img = synth_image(i, size, device, seed)                                          # resident in HBM before timing
        out.append({"image": img, "task": "detection", "input_ids": ids.to(idev), "attention_mask": mask.to(idev),
                    "positive_map_label_to_token": pmap, "is_thing": {c: (c <= 60) for c in pmap}})
    return out


def pin_host_threads(local_rank, world):
    """N ranks on one host: each rank gets its own slice of the usable cores (affinity by LOCAL_RANK) and sizes torch's intra-op pool
    to it, so 8 ranks on a 16-core box do not run 8 x 16 host threads (detectron2/engine/launch.py:67-126 leaves this to OMP_NUM_THREADS=1).
    Returns (cores of this rank, torch threads)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    if world > 1 and len(cores) >= world:
        per = len(cores) // world
        mine = cores[local_rank * per:(local_rank + 1) * per]
        try:
            os.sched_setaffinity(0, mine)
        except (AttributeError, OSError):
            pass
        cores = mine
    n = max(1, min(len(cores), 8))
    torch.set_num_threads(n)
    return len(cores), n


def synthetic_clip_tokenizer(context=77, vocab=49408):
    """stands in for open_clip.tokenize (its BPE vocabulary ships inside open_clip, which is not installable here): deterministic ids per
    string, start / end tokens where CLIP puts them -- the text tower's cost does not depend on which ids it sees."""
    import zlib

    def tok(texts):
        out = torch.zeros(len(texts), context, dtype=torch.long)
        for i, t in enumerate(texts):
            words = t.lower().replace(".", " .").split()[:context - 2]
            ids = [1 + zlib.crc32(w.encode()) % (vocab - 3) for w in words]
            out[i, 0] = vocab - 2
            out[i, 1:1 + len(ids)] = torch.tensor(ids, dtype=torch.long)
            out[i, 1 + len(ids)] = vocab - 1                      # end of text: the highest id of the row (CLIP.encode_text's argmax)
        return out
    return tok


def train_step_leg(dev, steps=3):
    """SURVEY row f-4, reported beside the headline: ONE TRAINING STEP (hipie_amd/training/step.py: the reference's coco_forward + three DINO
    criterion calls + the MaskDINO criterion + backward) at the reference's training batch -- ViT-H, 1024 x 1024, 2 images per GPU
    (configs/training/vit_huge_32g.yaml), the 80-class caption, 8 synthetic targets per image, DN_NUMBER 100, random-init weights.  Two
    untimed steps (allocator pools), then `steps` timed ones; then the same with gradient clipping and the AdamW step (`value`).  Never the
    headline; tools/bench_train_step.py is the forward + backward part of the same measurement."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    from bench_train_step import targets_for
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    from hipie_amd.training.step import TrainStep
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, Precision.parity(), device=dev)
    randomize_degenerate_inits(model)
    model.finalize()
    find = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False        # MIOpen's exhaustive search over the step's forward AND backward convolutions takes ~2 min
    for p in model.text_encoder.parameters():
        p.requires_grad_(False)
    B, size, L = 2, 1024, 194
    batch = synth_batch(cfg, B, size, 80, L, dev)
    targets = targets_for(B, 6, 2, size, L, dev)
    step = TrainStep(model)
    torch.cuda.reset_peak_memory_stats()
    fw, bw = [], []
    for it in range(steps + 2):
        model.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.enable_grad():
            losses = step.loss_dict(batch, targets)
            total = sum(losses.values())
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        total.backward()
        torch.cuda.synchronize()
        if it >= 2:
            fw.append(t1 - t0)
            bw.append(time.perf_counter() - t1)
    f, b = sum(fw) / len(fw), sum(bw) / len(bw)
    # the whole trainer iteration (SimpleTrainer.run_step: forward, backward, full-model gradient clipping at 0.1, AdamW with the 0.1 backbone
    # multiplier): one untimed iteration (the optimizer's state is allocated in it), then `steps` timed ones
    from hipie_amd.training.step import build_optimizer, train_iteration
    opt = build_optimizer(model)
    train_iteration(step, opt, batch, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        it_losses, norm = train_iteration(step, opt, batch, targets)
    torch.cuda.synchronize()
    it = (time.perf_counter() - t0) / steps
    res = {"value": round(B / it, 3), "unit": "images/sec per GPU", "ms_per_step": round(it * 1e3, 1), "forward_ms": round(f * 1e3, 1),
           "backward_ms": round(b * 1e3, 1), "steps": steps, "images_per_step": B,
           "loss_entries": len(losses), "finite": bool(torch.isfinite(total)) and bool(torch.isfinite(norm)),
           "peak_memory_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           "what": "one trainer iteration (hipie_amd/training/step.py::train_iteration: forward -> loss dictionary -> backward -> full-model gradient "
                   "clipping at 0.1 -> AdamW, multi-tensor on the device), ViT-H, 1024^2, 2 images per GPU, random-init weights, synthetic targets; "
                   "forward_ms / backward_ms from separately synchronised steps without the optimizer (their sum can exceed the free-running "
                   "iteration); the step is pinned against the reference's own coco_forward + criteria + backward by "
                   "tests/golden/train_step_tiny.npz"}
    del model, step, losses, total, opt, it_losses
    torch.cuda.empty_cache()
    torch.backends.cudnn.benchmark = find
    return res


def e2e_leg(model, batch, steps, dev):
    """`model(batched_inputs)` as the reference's evaluation timer sees it (detectron2/evaluation/evaluator.py:157-161): the images arrive
    as HOST tensors (uint8, what the dataset mapper produces), go to the device, and the call returns the FULL post-processed results
    (instances with masks, semantic and panoptic maps -- HIPIE_IMG.inference), plus MaskCLIP when model.enable_clip.  The upload of batch
    i + 1 runs on a copy stream (pinned memory, non-blocking) under the compute of batch i; `h2d_exposed_ms` = this figure minus the same
    loop with the images already resident."""
    host = [dict(b, image=b["image"].to(torch.uint8).cpu().pin_memory()) for b in batch]
    copy, main = torch.cuda.Stream(), torch.cuda.current_stream()

    def upload():
        with torch.cuda.stream(copy):
            imgs = [h["image"].to(dev, non_blocking=True) for h in host]
            ev = torch.cuda.Event()
            ev.record(copy)
        return imgs, ev

    def run(prefetch):
        nxt = upload() if prefetch else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if prefetch:
                imgs, ev = nxt
                main.wait_event(ev)
                for im in imgs:
                    im.record_stream(main)
                nxt = upload()
                b = [dict(h, image=im) for h, im in zip(host, imgs)]
            else:
                b = batch
            res = model(b)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, res
    run(True)                                              # warm-up of both forms (pinned buffers, allocator pools)
    t_res, _ = run(False)
    t_e2e, res = run(True)
    nbytes = sum(h["image"].numel() for h in host)
    return {"value": round(len(batch) / t_e2e, 3), "unit": "images/sec", "ms_per_step": round(t_e2e * 1e3, 2), "steps": steps,
            "ms_per_step_resident_inputs": round(t_res * 1e3, 2), "h2d_exposed_ms": round(max(t_e2e - t_res, 0.0) * 1e3, 2),
            "h2d_bytes_per_step": nbytes, "clip": bool(getattr(model, "enable_clip", False)),
            "results": sorted(res[0].keys()),
            "what": "model(batched_inputs): pinned uint8 host images -> double-buffered H2D on a copy stream -> forward -> HIPIE_IMG.inference "
                    "(instances + masks, sem_seg, panoptic_seg); the interval detectron2's evaluator times"}


def randomize_degenerate_inits(model):
    """default module init leaves rel_pos tables / MSDA offset weights / last bbox layers at zero, which would make the
    kernels' work trivial: redraw them N(0, 0.02) / a spread-out offset bias."""
    g = torch.Generator().manual_seed(0)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("rel_pos_h") or n.endswith("rel_pos_w") or "sampling_offsets.weight" in n or "attention_weights.weight" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)
            elif "sampling_offsets.bias" in n:
                p.copy_(torch.randn(p.shape, generator=g) * 1.5)
            elif p.dim() >= 2 and float(p.abs().sum()) == 0.0:
                p.copy_(torch.randn(p.shape, generator=g) * 0.02)


def _median_time(fn, runs=3, warmup=1):
    """SURVEY 8d protocol (mirrors detectron2/evaluation/evaluator.py:157-161): warm-up, then the median wall-clock of `runs`."""
    for _ in range(warmup):
        fn()
    ts = []
    for _ in range(runs):
        t0 = time.time()
        fn()
        ts.append(time.time() - t0)
    return sorted(ts)[len(ts) // 2]


def effective_cores():
    """host cores this process may actually use: os.cpu_count() limited by the affinity mask and a cgroup CPU quota (a container that
    sees 256 host cores but owns 32 runs SLOWER with 256 threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0:
                n = min(n, max(1, int(float(quota) / period)))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(cfg_dict, size, L, n_classes, budget_s=150.0, emit=None):
    """oracle (kind "port") on the host cores: ONE image of the timed batch through the WHOLE path (all 32 ViT-H blocks, text encoder,
    both heads) -- a bounded, un-extrapolated sample of the same workload.  `emit(dict)` is called with a preliminary SAMPLED figure
    (one windowed + one global block timed and scaled, heads on a 2-block ViT) before the full forward starts, so that a parent with a
    hard timeout always has a labelled number; the final figure replaces it."""
    from oracle import model as om
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    t_start = time.time()
    cores = effective_cores()
    threads = min(cores, 64)                 # torch's CPU kernels stop scaling (and then slow down) far below a 2-socket core count
    torch.set_num_threads(threads)
    torch.set_grad_enabled(False)
    note = "torch.set_num_threads(%d) [os.cpu_count() %d, usable %d]" % (threads, os.cpu_count() or 0, cores)
    c = dict(cfg_dict)
    nwin = len(cfg_dict["vit_window_blocks"])
    nglob = cfg_dict["vit_depth"] - nwin
    m = HIPIE_IMG(HipieConfig.from_dict(c), Precision.parity(), device="cpu")
    randomize_degenerate_inits(m)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    del m
    batch = synth_batch(None, 1, size, n_classes, L, "cpu", seed=1)
    ids, mask = batch[0]["input_ids"][None], batch[0]["attention_mask"][None]
    x = torch.randn(1, size // 16, size // 16, c["vit_embed_dim"])
    p = "detr.detr.backbone.0.backbone.blocks."
    wb = cfg_dict["vit_window_blocks"]
    gblk = [i for i in range(cfg_dict["vit_depth"]) if i not in wb][0]
    t_win = _median_time(lambda: om.vit_block(x, sd, p + "%d." % wb[0], c["vit_heads"], c["vit_window"]), runs=1)
    t_glob = _median_time(lambda: om.vit_block(x, sd, p + "%d." % gblk, c["vit_heads"], 0), runs=1)
    est_vit = nwin * t_win + nglob * t_glob
    prelim = {"value": round(1.0 / (2.2 * est_vit), 5), "unit": "images/sec", "cores": threads, "kind": "port", "extrapolated": True,
              "sample": "oracle/ (fp32 PyTorch CPU restatement of the reference), %s; PRELIMINARY, SAMPLED: one windowed (%.2f s) and one "
                        "global (%.2f s) ViT-H block of one %dx%d image scaled to %d + %d blocks, x 2.2 for the text encoder and the heads (the ratio of a full forward measured on an 8-core host) "
                        "-- reported only if the full forward did not finish inside the time bound" % (note, t_win, t_glob, size, size, nwin, nglob)}
    if emit is not None:
        emit(prelim)
    if (time.time() - t_start) + 2.3 * est_vit > budget_s:
        return prelim

    def full_forward():
        lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", c)
        om.coco_inference([batch[0]["image"]], lang, sd, c, task="detection")
    t0 = time.time()
    full_forward()
    s_img = time.time() - t0
    return {"value": round(1.0 / s_img, 5), "unit": "images/sec", "cores": threads, "kind": "port", "extrapolated": False,
            "sample": "oracle/ (fp32 PyTorch CPU restatement of the reference), %s: ONE full forward of one image %dx%d of the timed batch, "
                      "L=%d (all %d ViT-H blocks + text encoder + both heads): %.1f s  (single blocks: windowed %.2f s, global %.2f s)"
                      % (note, size, size, L, cfg_dict["vit_depth"], s_img, t_win, t_glob)}


def cpu_baseline_r50(size=512):
    """BASELINE configs[0] on the host cores ("the reference's CPU-runnable case"): the oracle's whole path behind the ResNet-50
    backbone on ONE 512 x 512 image with ONE referring expression, 1 warm-up + median of 3."""
    from oracle import model as om
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    threads = min(effective_cores(), 64)
    torch.set_num_threads(threads)
    torch.set_grad_enabled(False)
    cfg = HipieConfig.r50()
    c = cfg.to_dict()
    m = HIPIE_IMG(cfg, Precision.parity(), device="cpu")
    randomize_degenerate_inits(m)
    sd = {k: v.float() for k, v in m.state_dict().items()}
    del m
    batch = synth_batch(None, 1, size, 1, 12, "cpu", seed=1, task="grounding")
    ids, mask = batch[0]["input_ids"][None], batch[0]["attention_mask"][None]

    def full_forward():
        lang = om.bert_encoder(ids, mask, sd, "text_encoder.body.model.", c)
        om.coco_inference([batch[0]["image"]], lang, sd, c, task="grounding")
    t = _median_time(full_forward, runs=3, warmup=1)
    return {"value": round(1.0 / t, 4), "unit": "images/sec", "cores": threads, "kind": "port", "extrapolated": False,
            "sample": "BASELINE configs[0]: oracle/ (fp32 PyTorch CPU restatement of the reference) with the ResNet-50 backbone, ONE %dx%d image + "
                      "ONE text prompt (L=%d), full forward (text encoder, backbone, both heads): %.2f s, median of 3 after 1 warm-up, "
                      "torch.set_num_threads(%d)" % (size, size, int(ids.shape[1]), t, threads)}


def parity_error(policy, device, full_size=True):
    """max|a-b| / max|b| of every a22 output under `policy` against the fixtures produced by the reference's own
    DDETRSegmUniDN.coco_inference (top-k pinned): tests/golden/e2e_full.npz (the headline configuration itself: full ViT-H, shipped
    head sizes, one 1024 x 1024 image; big outputs on a strided subsample), e2e_deep.npz (the SHIPPED depths on a narrow ViT, two
    images) and e2e_tiny.npz -- the same checks as tests/test_gpu_e2e.py, run by the bench so that the timed arithmetic carries its
    own error."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _synth
    from util import Golden, rel_err
    from hipie_amd.config import HipieConfig
    from hipie_amd.hipie_img import HIPIE_IMG
    keys = ["pred_logits", "pred_boxes", "pred_boxious", "pred_masks", "reference_points", "pred_masks_maskdino", "pred_logits_maskdino",
            "pred_boxes_maskdino"]
    res, worst = {}, 0.0
    fixtures = ["e2e_deep", "e2e_tiny"]
    if full_size and os.path.exists(os.path.join(ROOT, "tests", "golden", "e2e_full_refinit.npz")):
        fixtures.insert(0, "e2e_full_refinit")      # the headline configuration with weights from the reference's OWN initialisation
    if full_size and os.path.exists(os.path.join(ROOT, "tests", "golden", "e2e_full.npz")):
        fixtures.insert(0, "e2e_full")              # ... and from the (harder) default synthetic distribution: the gate
    if full_size and os.path.exists(os.path.join(ROOT, "tests", "golden", "e2e_full_c80.npz")):
        fixtures.insert(0, "e2e_full_c80")          # ... and on the TIMED inputs themselves: image 0 of synth_batch, the 80-class caption (L = 194)
    for fixture in fixtures:                        #     and the grounding call (same gate weights as e2e_full)
        g = Golden(fixture)
        model = HIPIE_IMG(HipieConfig.from_dict(g.meta["cfg"]), policy, device=device)
        model.load_state_dict(_synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}, dist=g.meta.get("dist")), strict=True)
        if "bench_inputs" in g.meta:
            bi, per_task = g.meta["bench_inputs"], {}
            for task in ("detection", "grounding"):
                b = synth_batch(None, 1, bi["size"], bi["n_classes"], bi["L"], device, seed=bi["seed"], task=task)
                model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
                out = model.forward_raw(b)
                per_task[task] = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in keys}
            errs = {k: max(per_task[t][k] for t in per_task) for k in keys}
            res[fixture] = {"max": float("%.2e" % max(errs.values())), "per_output": {k: float("%.1e" % v) for k, v in errs.items()},
                            "per_task_max": {t: float("%.2e" % max(v.values())) for t, v in per_task.items()}}
            worst = max(worst, max(errs.values()))
            del model
            torch.cuda.empty_cache()
            continue
        imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
        ids, mask, pmap = _synth.synth_token_ids(2, g.meta["detection"]["n_classes"], 64, seed=74)
        model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
        out = model.forward_raw([{"image": im, "task": "detection", "input_ids": ids[i], "attention_mask": mask[i],
                                  "positive_map_label_to_token": pmap} for i, im in enumerate(imgs)])
        errs = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in keys}
        res[fixture] = {"max": float("%.2e" % max(errs.values())), "per_output": {k: float("%.1e" % v) for k, v in errs.items()}}
        worst = max(worst, max(errs.values()))
        del model
        torch.cuda.empty_cache()
    return {"max": float("%.2e" % worst), "tolerance": 1e-3, "within_tolerance": bool(worst <= 1e-3),
            "within_tolerance_on": [f for f in fixtures if res[f]["max"] <= 1e-3], "fixtures": res,
            "against": "tests/golden/{%s}.npz: the reference's own coco_inference on the CPU, pinned top-k (e2e_full = full ViT-H at "
                       "1024x1024 with the default synthetic weights -- the gate; e2e_full_refinit = the same with weights drawn from the "
                       "reference's own initialisation, tests/golden/refinit_stats.json; e2e_deep = the shipped depths on a narrow ViT; "
                       "e2e_full_c80 = the gate weights on the TIMED inputs: image 0 of this script's batch, the 80-class caption of 194 tokens, "
                       "and the grounding call)"
                       % ",".join(fixtures),
            "batch_note": "every fixture is a 1- or 2-image batch (the reference runs in minutes per image on the CPU); the timed batch is 8: the "
                          "bridge is tests/test_gpu_e2e.py::test_e2e_batch_items_are_independent (an image's rows do not depend on its batch "
                          "neighbours) and ::test_two_stream_step_repeats_bit_for_bit_at_full_size (the timed bs-8 step repeats within 7e-6)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--model", default="vit_huge", choices=["vit_huge", "vit_large", "vit_base", "r50"],
                    help="r50 = BASELINE configs[1] (use --batch 4); no ViT attention kernel there: roofline fields are null")
    ap.add_argument("--text-len", type=int, default=194, help="token length of the class prompt: 194 = the 80-class caption "
                    "as is (PAD_MAX off); 4096 = the shipped eval setting that pads every caption to 4096 tokens (SURVEY 8d)")
    ap.add_argument("--classes", type=int, default=80, help="vocabulary size of the class prompt: 80 (configs[2]); "
                    "--classes 150 --text-len 815 --size 1344 is the ADE-150 shape of configs[3] (chunked BERT, 84x84 grid), "
                    "--classes 1203 --text-len 4096 --size 1344 the LVIS shape of configs[4]")
    ap.add_argument("--task", default="detection", choices=["detection", "grounding"],
                    help="grounding = the referring-expression call of BASELINE configs[2] (one ~12-token expression)")
    ap.add_argument("--precision", default="split", choices=["split", "fast", "parity", "bf16", "default"],
                    help="split (default, the timed policy): split-fp16 GEMMs and attention logits, within 1e-3 of the reference at the "
                         "shipped depths; fast: single-fp16 operands (out of tolerance); parity: fp32 library GEMMs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-leg", action="store_true", help="skip parity_err and the fast-policy timing (rank 0, N = 1)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the training-step figure (`train_step`: SURVEY f-4; rank 0, N = 1, ~25 s)")
    ap.add_argument("--graph", action="store_true", help="capture the forward once into a hipGraph and replay it (same rate as "
                    "eager: the step is GPU-bound; the round-2 replay fault was torch.topk, replaced by hipie_topk -- DESIGN.md "
                    "section 9)")
    ap.add_argument("--no-gemm-table", action="store_true", help="do not load the committed TunableOp table "
                    "(hipie_amd/tuning/*.csv: the hipBLASLt solution picked per ViT-H linear shape; library plumbing)")
    ap.add_argument("--timed-only", action="store_true", help="stop right after the timed region (for kernel traces: no "
                    "extra roofline / post-processing passes at the end of the trace)")
    ap.add_argument("--config", type=int, default=None, choices=[0, 1, 2, 3, 4],
                    help="literal BASELINE.json configs[i] (overrides --model/--size/--batch/--classes/--text-len/--task): 0 = R50, one 512x512 "
                         "image + one text prompt; 1 = R50, 1024^2, bs 4, 80 classes; 2 = ViT-H, 1024^2, bs 8, 80 classes (the default, the "
                         "configuration the metric is quoted on); 3 = ViT-H, 1024^2, bs 8 per GPU (64 images at --gpus 8), 150 ADE prompts "
                         "(L=815: chunked BERT); 4 = ViT-H, 1344^2 canvas (1333 long edge), global bs 8 (8 // gpus per GPU), 1203 LVIS prompts cut at L=4096")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--breakdown", action="store_true", help="after the timed region, time every hand-written kernel class "
                    "and the main stages of one extra step (stderr)")
    args = ap.parse_args()
    if args.config is not None:
        preset = {0: dict(model="r50", size=512, batch=1, classes=1, text_len=12, task="grounding"),
                  1: dict(model="r50", size=1024, batch=4, classes=80, text_len=194, task="detection"),
                  2: dict(model="vit_huge", size=1024, batch=8, classes=80, text_len=194, task="detection"),
                  3: dict(model="vit_huge", size=1024, batch=8, classes=150, text_len=815, task="detection"),
                  4: dict(model="vit_huge", size=1344, batch=max(1, 8 // max(args.gpus, 1)), classes=1203, text_len=4096, task="detection")}[args.config]
        for k, v in preset.items():
            setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: re-launch under torch.distributed.run, one rank per GPU, loopback rendezvous
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))

    if args.cpu_baseline_only:          # child process of the cpu_baseline leg (bounded by a timeout in the parent)
        from hipie_amd.config import HipieConfig
        cfg = getattr(HipieConfig, args.model)()
        def emit(d):
            print("CPU_BASELINE " + json.dumps(d), flush=True)
        # the CPU figure of configs[0] (R50, one 512 x 512 image, one text prompt; seconds) first, then the ViT-H one
        try:
            print("CPU_BASELINE_R50 " + json.dumps(cpu_baseline_r50()), flush=True)
        except Exception as e:
            print("CPU_BASELINE_R50 " + json.dumps({"value": None, "error": repr(e)[:200]}), flush=True)
        emit(cpu_baseline(cfg.to_dict(), args.size, 194, 80, emit=emit))
        return

    # ONE JSON line on stdout and nothing else: libraries write there too (RCCL prints a version banner through C stdio when its
    # communicator comes up, flushed at exit, i.e. BEHIND the line).  From here on file descriptor 1 is stderr; the result lines go to
    # the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit_line(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())

    from hipie_amd import ops, parallel
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    from hipie_amd.postprocess import inference, inference_compact

    # N = 1 opens a ONE-rank "nccl" process group, so the step's all-gather and the timing all-reduce run through RCCL exactly as at
    # N > 1 (dp.backend says which backend really ran; null = the group could not be created and the collectives are identities)
    t_start = time.perf_counter()
    rank, world, local = parallel.init_from_env(single_rank_group=True)
    host_cores, host_threads = pin_host_threads(local, world)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d, or "
                         "run `python bench.py --gpus %d` without a WORLD_SIZE in the environment)" % (args.gpus, world, args.gpus, args.gpus))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)

    table = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipie_amd", "tuning", "tunableop_gfx950_vith_bs8.csv")
    if not args.no_gemm_table and os.path.exists(table):
        # read-only use of a committed table: known shapes get the tuned hipBLASLt solution, everything else the default
        import shutil
        import tempfile
        tmp_table = os.path.join(tempfile.gettempdir(), "hipie_tunableop_%d.csv" % os.getpid())
        shutil.copyfile(table, tmp_table)              # TunableOp may rewrite its file at exit: never the committed one
        torch.cuda.tunable.set_filename(tmp_table)
        torch.cuda.tunable.enable(True)
        torch.cuda.tunable.tuning_enable(False)

    cfg = getattr(HipieConfig, args.model)()
    prec = {"split": Precision.split3(), "fast": Precision.fast(), "parity": Precision.parity(), "bf16": Precision.bf16(),
            "default": Precision()}[args.precision]
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, prec, device=dev)
    randomize_degenerate_inits(model)
    model.finalize()
    L, n_classes = args.text_len, args.classes
    # ONE global batch of batch * world images (image i seeded by i), cut like detectron2's InferenceSampler: rank r owns a contiguous
    # index range (BASELINE configs[3] at --gpus 8: 64 images, 8 per rank)
    shard = parallel.shard_range(args.batch * world, rank, world)
    batch = synth_batch(cfg, len(shard), args.size, n_classes, L, dev, seed=0, task=args.task, indices=shard,
                        ids_device=dev if args.graph else "cpu")

    def local_step():
        out = model.forward_raw(batch)
        # the step ends at the compact per-image predictions that the data-parallel all-gather carries: class
        # scores, NMS, top-100, box scaling -- built on the device without a host round trip (postprocess.inference_compact ==
        # compact_predictions(inference(...)), tests/test_gpu_post.py); instance masks and the semantic / panoptic maps of the
        # full post-processing are timed separately below
        return inference_compact(model, out, batch, topk=100)

    def step():
        return parallel.all_gather_predictions(local_step())

    torch.cuda.synchronize()
    t_built = time.perf_counter()
    gathered = step()                                                      # first step: MIOpen find, TunableOp table, lazy HL8 weight copies
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    for _ in range(max(args.warmup, 1) - 1):
        gathered = step()
    dp = parallel.dp_evidence(gathered, args.batch, rank, world, dev)     # rccl_ranks (all-reduce of ones), shard, gathered shape
    # per-rank start-up cost (it sits inside the driver's wall clock at --gpus N): process start -> model built and finalised -> first step done
    dp["startup_s"] = {"build_max_over_ranks": round(parallel.max_over_ranks(t_built - t_start, dev), 2),
                       "first_step_max_over_ranks": round(parallel.max_over_ranks(t_first - t_built, dev), 2),
                       "host_cores_per_rank": host_cores, "torch_threads_per_rank": host_threads}

    # The forward has static shapes and no host<->device traffic, so it CAN be captured once into a hipGraph and replayed
    # (--graph).  Default is eager: with batched post-processing and the fused glue kernels the step is GPU-bound.
    graph, gblock = None, None
    if args.graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                model.forward_raw(batch)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gout = model.forward_raw(batch)          # the forward only: the post-processing syncs once for its counts
            torch.cuda.synchronize()

            def step():  # noqa: F811
                graph.replay()
                return parallel.all_gather_predictions(inference_compact(model, gout, batch, topk=100))
            step()
            torch.cuda.synchronize()
        except Exception as e:          # fall back to eager launches, say so in the JSON line
            print("bench: hipGraph capture failed (%r); running eagerly" % (e,), file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

            def step():  # noqa: F811
                return parallel.all_gather_predictions(local_step())

    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt, dev)

    if args.timed_only:
        if rank == 0:
            emit_line({"ms_per_step": round(dt / args.steps * 1e3, 2), "images_per_sec": round(args.batch * world * args.steps / dt, 3)})
        parallel.shutdown()
        return

    # dominant hand-written kernel, timed live with HIP events on the launch stream: the same 24 launches per step of the
    # global-attention kernel, issued eagerly (events cannot be placed inside a graph replay) right after the timed region
    gemm_tags = ("gemm_qkv", "gemm_proj", "gemm_fc1", "gemm_fc2")
    ops.PROFILE.enable("vit_attn_global", *gemm_tags)
    for _ in range(min(args.steps, 3)):
        model.forward_raw(batch)
    kern_ms, kern_n = ops.PROFILE.mean_ms("vit_attn_global")
    gemm = {t: ops.PROFILE.mean_ms(t) for t in gemm_tags}
    ops.PROFILE.disable()

    # the shipped eval setting: MODEL.LANGUAGE_BACKBONE.PAD_MAX with MAX_QUERY_LEN 4096 -- the same batch, the caption padded to
    # 4096 tokens; same step definition, same policy.  The padding rows are zero behind the reference's > 512 text branch, so the
    # product computes the real tokens plus one padding row and repeats its class-logit column (tests/test_gpu_e2e.py::test_e2e_pad_max_4096_is_trimmed)
    pad_max = None
    if rank == 0 and world == 1 and n_classes == 80 and L < 4096 and args.task == "detection" and graph is None and not args.no_parity_leg:
        pbatch = synth_batch(cfg, len(shard), args.size, n_classes, 4096, dev, seed=0, task=args.task, indices=shard)
        for b_, p_ in zip(batch, pbatch):
            p_["image"] = b_["image"]                  # the same resident images

        def padstep():
            return inference_compact(model, model.forward_raw(pbatch), pbatch, topk=100)
        for _ in range(2):
            padstep()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            padstep()
        torch.cuda.synchronize()
        pdt_ = (time.perf_counter() - t1) / args.steps
        pad_max = {"text_len": 4096, "attended_tokens": int(pbatch[0]["attention_mask"].sum()), "ms_per_step": round(pdt_ * 1e3, 2),
                   "value": round(args.batch / pdt_, 3), "unit": "images/sec", "steps": args.steps,
                   "pred_logits_shape": None, "vs_unpadded_step": round(pdt_ / (dt / args.steps), 4)}
        po = model.forward_raw(pbatch)
        pad_max["pred_logits_shape"] = list(po["pred_logits"].shape)
        del po, pbatch

    # the full post-processing of the reference's eval branch (instance masks at 1024^2, semantic + panoptic maps),
    # timed on its own: it is the row after the a22 metric surface
    post_ms = None
    if rank == 0:
        out = model.forward_raw(batch)
        inference(model, out, batch)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        inference(model, out, batch)
        torch.cuda.synchronize()
        post_ms = (time.perf_counter() - t1) * 1e3
        del out

    # secondary figures (rank 0, N = 1): the end-to-end call as the reference's timer sees it, and the same with MaskCLIP at the size the
    # eval yamls run it (MODEL.CLIP.NAME ViT-L-14-336, random weights, a synthetic tokenizer: open_clip's BPE vocabulary is not available here)
    e2e = e2e_clip = post_clip_ms = None
    if rank == 0 and world == 1 and args.task == "detection" and graph is None and not args.no_parity_leg:
        try:
            e2e = e2e_leg(model, batch, max(2, min(args.steps, 5)), dev)
            from hipie_amd.modeling.transformer import set_split
            from hipie_amd.open_vocab import MaskCLIP
            torch.manual_seed(1)
            model.clip = MaskCLIP("ViT-L-14-336", tokenize=synthetic_clip_tokenizer()).to(dev).eval()
            model.clip.loaded = True
            set_split(model.clip, bool(prec.split))
            model.train_labels = [{"id": i, "name": "train%d,alias%d" % (i, i)} for i in range(133)]
            n_names = len(batch[0]["positive_map_label_to_token"])            # the classes whose names fit the caption (test vocabulary = the prompt's)
            names = [{"id": i, "name": ("train%d" % i) if i % 2 else ("novel%d" % i)} for i in range(n_names)]
            cbatch = [dict(b, open_seg_labels=names) for b in batch]
            model.enable_clip = True
            out = model.forward_raw(cbatch)
            inference(model, out, cbatch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            inference(model, out, cbatch)
            torch.cuda.synchronize()
            post_clip_ms = (time.perf_counter() - t1) * 1e3
            del out
            e2e_clip = e2e_leg(model, cbatch, 2, dev)
        except Exception as e:          # never lose the measured line to the side legs
            print("bench: e2e / clip leg failed: %r" % (e,), file=sys.stderr)
        finally:
            model.enable_clip = False
            model.clip = None
            torch.cuda.empty_cache()

    if args.breakdown and rank == 0:
        ops.PROFILE.enable("all")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        local_step()                     # no collective here: only rank 0 runs the breakdown
        torch.cuda.synchronize()
        print("BREAKDOWN step %.1f ms" % ((time.perf_counter() - t1) * 1e3), file=sys.stderr)
        for tag, (mean, n, tot) in sorted(ops.PROFILE.summary().items()):
            print("BREAKDOWN %-18s n=%4d mean=%8.3f ms total=%8.2f ms" % (tag, n, mean, tot), file=sys.stderr)
        ops.PROFILE.disable()

    # the timed arithmetic carries its own error (both reference-generated fixtures); the out-of-tolerance fp16 mode is timed beside
    # it for reference: same model, same batch, same step definition (N = 1, rank 0 only)
    parity_err = other = mixed = None
    if rank == 0 and world == 1 and not args.no_parity_leg:
        try:
            parity_err = parity_error(prec, dev)
            if args.precision != "fast" and args.model == "vit_huge":
                del model
                torch.cuda.empty_cache()
                torch.manual_seed(0)
                pm = HIPIE_IMG(cfg, Precision.fast(), device=dev)
                randomize_degenerate_inits(pm)
                pm.finalize()

                def pstep():
                    return inference_compact(pm, pm.forward_raw(batch), batch, topk=100)
                for _ in range(2):
                    pstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    pstep()
                torch.cuda.synchronize()
                pdt = (time.perf_counter() - t1) / args.steps
                other = {"precision_policy": "fast", "dtype": "f16", "value": round(args.batch / pdt, 3), "unit": "images/sec",
                         "ms_per_step": round(pdt * 1e3, 2), "steps": args.steps, "parity_err": parity_error(Precision.fast(), dev, full_size=False),
                         "note": "single-fp16 operands everywhere: OUTSIDE the 1e-3 tolerance, not the headline"}
                # the `mixed` policy: split linears + single-fp16 ViT attention core.  In tolerance with weights from the reference's own
                # initialisation, out of tolerance on the harder default synthetic distribution -- both reported, never the headline
                del pm
                torch.cuda.empty_cache()
                torch.manual_seed(0)
                mm = HIPIE_IMG(cfg, Precision.mixed(), device=dev)
                randomize_degenerate_inits(mm)
                mm.finalize()

                def mstep():
                    return inference_compact(mm, mm.forward_raw(batch), batch, topk=100)
                for _ in range(2):
                    mstep()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    mstep()
                torch.cuda.synchronize()
                mdt = (time.perf_counter() - t1) / args.steps
                del mm
                torch.cuda.empty_cache()
                mixed = {"precision_policy": "mixed", "dtype": "f16x3 linears + f16 ViT attention core", "value": round(args.batch / mdt, 3),
                         "unit": "images/sec", "ms_per_step": round(mdt * 1e3, 2), "steps": args.steps,
                         "parity_err": parity_error(Precision.mixed(), dev),
                         "note": "NOT the headline: within 1e-3 only on the fixtures listed in parity_err.within_tolerance_on (the reference's own "
                                 "initialisation distribution), not on the harder default synthetic weights that gate the timed policy"}
                model = None
        except Exception as e:          # never lose the measured line to the side legs
            print("bench: parity leg failed: %r" % (e,), file=sys.stderr)

    train_leg = None
    if rank == 0 and world == 1 and not args.no_train_leg and not args.timed_only and args.model == "vit_huge" and graph is None:
        try:
            model = None
            torch.cuda.empty_cache()
            train_leg = train_step_leg(dev)
        except Exception as e:          # never lose the measured line to the side legs
            print("bench: training-step leg failed: %r" % (e,), file=sys.stderr)

    if rank == 0:
        std = args.model == "vit_huge" and args.batch == 8 and args.size == 1024
        pmc = {}
        for pmc_file in ("r03_pmc_kernels.json", "r04_pmc_kernels.json", "r05_pmc_kernels.json", "r06_pmc_kernels.json"):      # the newest round's rows win
            pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_file)
            if os.path.exists(pmc_path) and std:
                pmc.update(json.load(open(pmc_path)))
        images = args.batch * world * args.steps
        N = (args.size // 16) ** 2
        E = cfg.vit_embed_dim
        flops = 4.0 * N * N * E * args.batch          # QK^T + PV of one global block, all heads, this batch
        ach = flops / (kern_ms * 1e-3) / 1e12 if kern_ms else None
        split = args.precision == "split"
        # the four ViT linears: algorithmic flops 2 M N K per launch (M = batch * tokens); the split form issues 3 MFMA flops per
        # algorithmic flop (x_lo.w_hi + x_hi.w_lo + x_hi.w_hi)
        M_tok = args.batch * N
        shapes = {"gemm_qkv": (E, 3 * E), "gemm_proj": (E, E), "gemm_fc1": (E, 4 * E), "gemm_fc2": (4 * E, E)}
        g_ms = sum(gemm[t][0] * gemm[t][1] for t in shapes if gemm.get(t, (None, 0))[0])
        g_n = sum(gemm[t][1] for t in shapes if gemm.get(t, (None, 0))[0])
        g_flop = sum(2.0 * M_tok * k * n * gemm[t][1] for t, (k, n) in shapes.items() if gemm.get(t, (None, 0))[0])
        g_ach = g_flop / (g_ms * 1e-3) / 1e12 if g_ms else None
        roof_attn = {"bound": "mfma", "kernel": "%s (ViT global attention, %d launches timed)" %
                     ("vit_attn_split_kernel<hd%d,NB2,8 waves>: split-fp16 logits (3 products), P and V as fp16 pairs (3 products); hd 80 = two 32-row blocks + a "
                      "16-row tail on v_mfma_f32_16x16x32_f16 (no head-dim padding)" % (E // cfg.vit_heads)
                      if split else "vit_attn_sp_kernel<%s,hd%d>" % (str(prec.attn).split(".")[-1], E // cfg.vit_heads), kern_n),
                     "achieved": None if ach is None else round(ach, 1), "peak": 2500.0, "unit": "TFLOP/s",
                     "frac": None if ach is None else round(ach / 2500.0, 4), "mfma_flops_per_algorithmic_flop": 3.0 if split else 1.0,
                     "frac_mfma_issued": None if ach is None else round(ach * (3.0 if split else 1.0) / 2500.0, 4),
                     "avg_launch_ms": None if not kern_ms else round(kern_ms, 4), "flop_per_launch": flops,
                     "traffic": pmc.get("attn_split", {}).get("traffic_bytes_per_launch") if split else None}
        if g_ach is not None:
            roof = {"bound": "mfma", "kernel": "gemm_kernel<320,split> (hipie_gemm, HL8 operands: the ViT qkv / proj / fc1 / fc2 linears, %d launches "
                                               "timed)" % g_n,
                    "achieved": round(g_ach, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(g_ach / 2500.0, 4),
                    "mfma_flops_per_algorithmic_flop": 3.0, "frac_mfma_issued": round(3.0 * g_ach / 2500.0, 4),
                    "note": "achieved = ALGORITHMIC flops (2 M N K) / time; the fp32-class product costs three fp16 MFMAs, so a frac of 1/3 "
                            "would be the matrix pipe saturated; peak = dense fp16 MFMA (gfx950 has no faster exact-fp32 path: fp32 MFMA "
                            "peaks at 157 TFLOP/s)",
                    "avg_launch_ms": round(g_ms / g_n, 4), "flop_per_launch": g_flop / g_n,
                    "per_shape_ms": {t: round(gemm[t][0], 4) for t in shapes if gemm.get(t, (None, 0))[0]},
                    "traffic": pmc.get("gemm_qkv_split", {}).get("traffic_bytes_per_launch"),
                    "traffic_note": "bytes per launch of the qkv shape (0.81 ms; 691 MB algorithmic), rocprofv3 PMC FETCH_SIZE (x2, guide correction) + "
                                    "WRITE_SIZE, separate passes: profiles/r06_pmc_kernels.json (the newest round present wins)"}
        else:
            roof = roof_attn
        line = {
            "metric": "images/sec @%dx%d %s bs=%d (single-image inference hot path: box, class and mask logits)"
                      % (args.size, args.size, {"vit_huge": "ViT-H", "vit_large": "ViT-L", "vit_base": "ViT-B", "r50": "R50"}[args.model], args.batch),
            "value": round(images / dt, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"split": "f16x3 (split-fp16 operands, fp32 accumulate: fp32-class)", "fast": "f16", "parity": "f32+f16attn", "bf16": "bf16",
                      "default": "bf16+f32head"}[args.precision],
            "data": "synthetic (uint8-valued random images resident in HBM BEFORE the timed region -- the 12.6 MB / image host-to-device copy that the "
                    "reference's timer includes (detectron2/evaluation/evaluator.py:157-161) is outside the step; synthetic BERT token ids "
                    + ("resident on the device (hipGraph capture)" if args.graph else "handed over as host tensors like a tokenizer's output")
                    + "; random-init weights)",
            "config": {"workload": "BASELINE.json configs[%s]: %s, %dx%d, batch %d per GPU, %d %s (L=%d), %s"
                                   % (str(args.config) if args.config is not None else
                                      {80: "1" if args.model == "r50" else ("2" if world == 1 else "2 per GPU (weak scaling of the metric's workload: "
                                            "%d images, 80-class prompt; configs[3] literally = --config 3)" % (args.batch * world)), 150: "3",
                                       1203: "4"}.get(n_classes, "-"),
                                      args.model, args.size, args.size, args.batch, n_classes,
                                      "text prompt" if args.task == "grounding" else "class prompts", L, args.task),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "precision_policy": args.precision,
                       "launch": "hipGraph replay" if graph is not None else "eager",
                       "constants": "weight- and geometry-only tensors (rel-pos tables, position embeddings, valid ratios) are "
                                    "built once; nothing that depends on image or text content is cached"},
            "dp": dp,
            "roofline": roof,
            "roofline_attention": roof_attn,
            "postprocess_full_ms": None if post_ms is None else round(post_ms, 2),
            "postprocess_clip_ms": None if post_clip_ms is None else round(post_clip_ms, 2),
            "value_e2e": e2e,
            "value_e2e_clip": e2e_clip,
            "train_step": train_leg,
            "parity_err": parity_err,
            "pad_max_4096": pad_max,
            "fast_policy": other,
            "mixed_policy": mixed,
        }
        # SURVEY 8d names the measurement weights: default module init with the zero tensors redrawn = tests/golden/e2e_full_refinit.  `value`
        # stays the policy that ALSO passes the (much harder) default synthetic gate; value_8d says what the contract's own distribution allows:
        # the fastest policy measured here whose error on e2e_full_refinit is <= 5e-4, with its error on the gate fixture beside it
        cands = [("split", line["value"], line["ms_per_step"], parity_err)] if parity_err else []
        if mixed:
            cands.append(("mixed", mixed["value"], mixed["ms_per_step"], mixed["parity_err"]))
        if other and "e2e_full_refinit" in (other.get("parity_err") or {}).get("fixtures", {}):
            cands.append(("fast", other["value"], other["ms_per_step"], other["parity_err"]))
        ok = [c for c in cands if c[3] and c[3]["fixtures"].get("e2e_full_refinit", {}).get("max", 1.0) <= 5e-4]
        if ok:
            best = max(ok, key=lambda c: c[1])
            line["value_8d"] = {"value": best[1], "unit": "images/sec", "ms_per_step": best[2], "precision_policy": best[0],
                                "err_on_e2e_full_refinit": best[3]["fixtures"]["e2e_full_refinit"]["max"],
                                "err_on_gate_e2e_full": best[3]["fixtures"].get("e2e_full", {}).get("max"),
                                "note": "fastest measured policy within 5e-4 on the weights SURVEY 8d names (reference init, zero tensors redrawn); "
                                        "NOT the headline -- `value` is the policy that also passes the harder synthetic gate"}
        if not args.no_cpu_baseline and world == 1 and args.model != "r50" and args.classes == 80:   # rank 0 at N = 1 only (bounded ~20 s CPU sample)
            import subprocess
            try:                        # separate process, hard time bound: the baseline must never break the measured line
                env = dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
                for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                    env.pop(k, None)
                cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--model", args.model, "--size", str(args.size)]
                try:
                    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
                    out_txt, err_txt = r.stdout, r.stderr
                except subprocess.TimeoutExpired as te:          # keep what the child had printed: its preliminary, labelled figure
                    out_txt = te.stdout.decode() if isinstance(te.stdout, bytes) else (te.stdout or "")
                    err_txt = "timeout after 240 s"
                tag = [l for l in out_txt.splitlines() if l.startswith("CPU_BASELINE ")]
                line["cpu_baseline"] = json.loads(tag[-1][len("CPU_BASELINE "):]) if tag else \
                    {"value": None, "error": (err_txt or out_txt)[-300:]}
                tag = [l for l in out_txt.splitlines() if l.startswith("CPU_BASELINE_R50 ")]
                if tag:
                    line["cpu_baseline_r50_512"] = json.loads(tag[-1][len("CPU_BASELINE_R50 "):])
            except Exception as e:
                line["cpu_baseline"] = {"value": None, "error": repr(e)[:200]}
        emit_line(line)
    parallel.barrier()
    parallel.shutdown()


if __name__ == "__main__":
    main()
