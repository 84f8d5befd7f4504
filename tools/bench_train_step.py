#!/usr/bin/env python3
"""forward + backward time of ONE TRAINING STEP (hipie_amd/training/step.py) at the reference's training batch: ViT-H, 1024 x 1024, 2 images per
GPU (configs/training/vit_huge_32g.yaml:1 -- 32 GPUs x 2), the 80-class caption, 8 synthetic targets per image (6 things, 2 stuff), DN_NUMBER 100,
12544 mask points, random-init weights.  Prints ms for the forward (loss dictionary), the backward, and the peak memory.
    python tools/bench_train_step.py [batch] [steps]
env: LIB_LINEAR=1 (library fp32 linears), PHASES=1 (synchronised forward phases), TORCH_PROF=1 (top kernels), HOSTPROF=1 (cProfile of a forward)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from hipie_amd.config import HipieConfig, Precision  # noqa: E402
from hipie_amd.hipie_img import HIPIE_IMG  # noqa: E402
from hipie_amd.training.step import TrainStep  # noqa: E402


def targets_for(batch, n_things, n_stuff, size, L, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(batch):
        n = n_things + n_stuff
        c = torch.rand(n, 2, generator=g) * 0.6 + 0.2
        s = torch.rand(n, 2, generator=g) * 0.3 + 0.05
        pm = torch.zeros(n, L, dtype=torch.bool)
        for t in range(n):
            a = int(torch.randint(1, L - 3, (1,), generator=g))
            pm[t, a:a + 2] = True
        thing = torch.ones(n, dtype=torch.bool)
        thing[n_things:] = False
        masks = torch.zeros(n, size, size)
        for t in range(n):
            x0, y0 = int((c[t, 0] - s[t, 0] / 2) * size), int((c[t, 1] - s[t, 1] / 2) * size)
            masks[t, y0:y0 + max(4, int(s[t, 1] * size)), x0:x0 + max(4, int(s[t, 0] * size))] = 1
        out.append({"labels": torch.randint(0, 80, (n,), generator=g).to(dev), "boxes": torch.cat((c, s), 1).to(dev), "positive_map": pm.to(dev),
                    "is_thing": thing.to(dev), "masks": masks.to(dev), "image_size": torch.tensor([size, size, size, size], dtype=torch.float, device=dev)})
    return out


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dev = torch.device("cuda", 0)
    cfg = HipieConfig.vit_huge()
    torch.manual_seed(0)
    model = HIPIE_IMG(cfg, Precision.parity(), device=dev)
    bench.randomize_degenerate_inits(model)
    model.finalize()
    for p in model.text_encoder.parameters():
        p.requires_grad_(False)
    size, L = 1024, 194
    batch = bench.synth_batch(cfg, B, size, 80, L, dev)
    targets = targets_for(B, 6, 2, size, L, dev)
    step = TrainStep(model)
    from hipie_amd.training import net
    if os.environ.get("LIB_LINEAR") == "1":                  # A/B: the big linears on the library instead of the split GEMM Function

        class LibBackend(net.HipBackend):
            linear = None
        step.be = LibBackend
    phases = {}
    if os.environ.get("PHASES") == "1":                      # synchronised wall time of the forward's phases
        def wrap(mod, name):
            f = getattr(mod, name)

            def g(*a, **k):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r = f(*a, **k)
                torch.cuda.synchronize()
                phases[name] = phases.get(name, 0.0) + time.perf_counter() - t0
                return r
            setattr(mod, name, g)
        for nm in ("backbone_and_projections", "hipie_transformer", "maskdino_pixel_decoder", "maskdino_decoder"):
            wrap(net, nm)
        wrap(step, "maskdino_losses")
        wrap(step.criterion, "forward")
    n_par = sum(p.numel() for p in model.parameters() if p.requires_grad)
    fw, bw = [], []
    for it in range(steps + 2):                              # two untimed steps: the second still grows the allocator's pools (2.1 s forward)
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
        t2 = time.perf_counter()
        if it >= 2:
            fw.append(t1 - t0)
            bw.append(t2 - t1)
    if os.environ.get("TORCH_PROF") == "1":                  # top kernels of one steady-state step
        from torch.profiler import ProfilerActivity, profile
        model.zero_grad(set_to_none=True)
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            with torch.enable_grad():
                total = sum(step.loss_dict(batch, targets).values())
            total.backward()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=90))
    if os.environ.get("HOSTPROF") == "1":                    # where the HOST time goes (the step is launch-bound on a slow host)
        import cProfile
        import io
        import pstats
        model.zero_grad(set_to_none=True)
        pr = cProfile.Profile()
        pr.enable()
        with torch.enable_grad():
            total = sum(step.loss_dict(batch, targets).values())
        pr.disable()
        torch.cuda.synchronize()
        for key in ("cumulative", "tottime"):
            out = io.StringIO()
            pstats.Stats(pr, stream=out).sort_stats(key).print_stats(28)
            print("\n".join(l[:190] for l in out.getvalue().splitlines() if l.strip()))
        total.backward()
        torch.cuda.synchronize()
    if os.environ.get("HOSTPROF") == "2":                    # wall-clock stack sampling of the main thread (1 ms) over 3 forwards: where the host IS
        import collections
        import threading
        main_id = threading.get_ident()
        hits, leaf, stop = collections.Counter(), collections.Counter(), threading.Event()

        def sampler():
            while not stop.is_set():
                f = sys._current_frames().get(main_id)
                chain, lines = [], []
                while f is not None:
                    fn = f.f_code.co_filename
                    if "/hipie_amd/" in fn:
                        chain.append("%s:%s" % (os.path.basename(fn), f.f_code.co_name))
                        lines.append(f.f_lineno)
                    f = f.f_back
                if chain:
                    leaf[chain[0] + ":%d" % lines[0]] += 1
                    for c in set(chain):
                        hits[c] += 1
                time.sleep(0.001)
        import scipy.optimize as _m
        lsa, lsa_t = _m.linear_sum_assignment, []

        def timed_lsa(c):
            t0_ = time.perf_counter()
            r = lsa(c)
            lsa_t.append((time.perf_counter() - t0_, c.shape))
            return r
        _m.linear_sum_assignment = timed_lsa
        for _ in range(3):
            model.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            th = threading.Thread(target=sampler)
            stop.clear()
            th.start()
            with torch.enable_grad():
                total = sum(step.loss_dict(batch, targets).values())
            stop.set()
            th.join()
            total.backward()
            torch.cuda.synchronize()
        _m.linear_sum_assignment = lsa
        print("scipy linear_sum_assignment: %d calls, %.1f ms in total over 3 forwards; slowest %s" % (
            len(lsa_t), 1e3 * sum(t for t, _ in lsa_t), ["%.1f ms %s" % (1e3 * t, sh) for t, sh in sorted(lsa_t, reverse=True)[:4]]))
        n = sum(leaf.values())
        print("host samples over 3 forwards: %d (~ms); innermost hipie_amd frame:" % n)
        for k, v in leaf.most_common(22):
            print("   %5d  %s" % (v, k))
        print("inclusive:")
        for k, v in hits.most_common(30):
            print("   %5d  %s" % (v, k))
    if phases:
        print("forward phases (sum over %d steps, ms): %s" % (steps + 1, ", ".join("%s %.1f" % (k, v * 1e3) for k, v in phases.items())))
    ms_ = torch.cuda.memory_stats()
    print("allocator: %d device allocations, %d device frees, %d retries over %d steps; reserved %.1f GB" % (
        ms_.get("num_device_alloc", -1), ms_.get("num_device_free", -1), ms_.get("num_alloc_retries", -1), steps + 2, torch.cuda.memory_reserved() / 2 ** 30))
    print("per step (ms) forward: %s | backward: %s" % (" ".join("%.0f" % (1e3 * t) for t in fw), " ".join("%.0f" % (1e3 * t) for t in bw)))
    gn = float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in model.parameters() if p.grad is not None)))
    print("training step, ViT-H 1024^2, %d images / GPU, %.0f M trainable parameters, %d loss entries: forward %.1f ms, backward %.1f ms, total %.1f ms "
          "(%.2f images/s per GPU); loss %.3f, gradient norm %.3e, finite %s; peak memory %.1f GB"
          % (B, n_par / 1e6, len(losses), 1e3 * sum(fw) / len(fw), 1e3 * sum(bw) / len(bw), 1e3 * (sum(fw) + sum(bw)) / len(fw),
             B * len(fw) / (sum(fw) + sum(bw)), float(total), gn, bool(torch.isfinite(total)) and gn == gn, torch.cuda.max_memory_allocated() / 2 ** 30))


if __name__ == "__main__":
    main()
