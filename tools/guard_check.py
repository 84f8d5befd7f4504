#!/usr/bin/env python3
"""Out-of-bounds WRITE detector for the hand-written kernels: every tensor the op layer (hipie_amd/ops.py) allocates is placed between two
4 KiB guard zones filled with 0xA5; after every op the device is synchronised and the guards of the tensors that op allocated are checked.
Runs the timed workload's forward (ViT-H, bs 8, 1024 x 1024) or, with `tiny`, the small e2e fixture model.  usage: guard_check.py [tiny]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from hipie_amd import ops  # noqa: E402

G = 4096
LIVE = []           # (buffer, nbytes) of the op in flight
BAD = []


class TorchProxy(object):
    def __getattr__(self, name):
        return getattr(torch, name)

    @staticmethod
    def _guarded(shape, dtype, device, zero=False):
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        pad = (-nbytes) % 256
        buf = torch.full((G + nbytes + pad + G,), 0xA5, dtype=torch.uint8, device=device)
        body = buf[G:G + nbytes]
        if zero:
            body.zero_()
        LIVE.append((buf, nbytes))
        return body.view(dtype).view(*shape) if n else torch.empty(shape, dtype=dtype, device=device)

    def empty(self, *shape, dtype=torch.float32, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        if device is None or torch.device(device).type != "cuda":
            return torch.empty(*shape, dtype=dtype, device=device, **kw)
        return self._guarded(shape, dtype, device)

    def zeros(self, *shape, dtype=torch.float32, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        if device is None or torch.device(device).type != "cuda":
            return torch.zeros(*shape, dtype=dtype, device=device, **kw)
        return self._guarded(shape, dtype, device, zero=True)

    def empty_like(self, x, dtype=None, memory_format=None, **kw):
        if not x.is_cuda or (memory_format is None and not x.is_contiguous()):
            return torch.empty_like(x, dtype=dtype, **kw) if memory_format is None else torch.empty_like(x, dtype=dtype, memory_format=memory_format, **kw)
        return self._guarded(tuple(x.shape), dtype or x.dtype, x.device)


def check(name):
    torch.cuda.synchronize()
    for buf, nbytes in LIVE:
        lo, hi = buf[:G], buf[G + nbytes:]
        if bool((lo != 0xA5).any()) or bool((hi != 0xA5).any()):
            nlo, nhi = int((lo != 0xA5).sum()), int((hi != 0xA5).sum())
            first_hi = int(torch.nonzero(hi != 0xA5)[0]) if nhi else -1
            BAD.append((name, nbytes, nlo, nhi, first_hi))
            print("GUARD VIOLATION in %s: tensor of %d bytes, %d bytes stomped below, %d above (first at +%d)" % (name, nbytes, nlo, nhi, first_hi), flush=True)
    LIVE.clear()


def install():
    ops.torch = TorchProxy()
    for name in dir(ops):
        fn = getattr(ops, name)
        if callable(fn) and getattr(fn, "__module__", None) == ops.__name__ and not name.startswith("_") and name not in ("hl8_pack", "hl8_unpack", "split_weight"):
            def make(fn, name):
                def wrapped(*a, **k):
                    depth[0] += 1
                    try:
                        out = fn(*a, **k)
                    finally:
                        depth[0] -= 1
                    if depth[0] == 0:
                        check(name)
                    return out
                wrapped.__name__ = name
                return wrapped
            setattr(ops, name, make(fn, name))


depth = [0]


def main():
    torch.set_grad_enabled(False)
    install()
    import bench
    from hipie_amd.config import HipieConfig, Precision
    from hipie_amd.hipie_img import HIPIE_IMG
    dev = torch.device("cuda", 0)
    if len(sys.argv) > 1 and sys.argv[1] == "tiny":
        import _synth
        from util import Golden
        g = Golden("e2e_tiny")
        model = HIPIE_IMG(HipieConfig.from_dict(g.meta["cfg"]), Precision.split3(), device=dev)
        model.load_state_dict(_synth.synth_full_state_dict({k: tuple(v) for k, v in g.meta["manifest"].items()}), strict=True)
        imgs = _synth.synth_images([tuple(s) for s in g.meta["sizes"]], seed=73)
        ids, mask, pmap = _synth.synth_token_ids(2, 9, 64, seed=74)
        batch = [{"image": im, "task": "detection", "input_ids": ids[i], "attention_mask": mask[i]} for i, im in enumerate(imgs)]
    else:
        cfg = HipieConfig.vit_huge()
        torch.manual_seed(0)
        model = HIPIE_IMG(cfg, Precision.split3(), device=dev)
        bench.randomize_degenerate_inits(model)
        batch = bench.synth_batch(cfg, 8, 1024, 80, 194, dev)
    model.finalize()
    for it in range(2):
        model.forward_raw(batch)
        torch.cuda.synchronize()
        print("forward %d done, %d violations so far" % (it, len(BAD)), flush=True)
    print("GUARD CHECK: %s" % ("clean" if not BAD else "%d violations: %s" % (len(BAD), sorted(set(b[0] for b in BAD)))))


if __name__ == "__main__":
    main()
