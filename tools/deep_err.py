#!/usr/bin/env python3
"""a22 errors of the precision policies on the full-depth fixture (tests/golden/e2e_deep.npz) and the tiny ones, on the GPU.
    python tools/deep_err.py [policy ...]      (default: parity fast)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("HIPIE_MIOPEN_FIND", "0")
from test_gpu_e2e import KEYS, build, inputs  # noqa: E402
from util import rel_err  # noqa: E402

torch.set_grad_enabled(False)


def main():
    from hipie_amd.config import Precision
    names = [a for a in sys.argv[1:] if not a.startswith("fixtures=")] or ["parity", "fast"]
    fixtures = ([a[9:].split(",") for a in sys.argv[1:] if a.startswith("fixtures=")] or [["e2e_tiny", "e2e_deep"]])[0]
    for fixture in fixtures:
        print("%-18s " % fixture + " ".join("%-9s" % k.replace("pred_", "").replace("maskdino", "md")[:9] for k in KEYS) + "  max")
        for n in names:
            g, model = build(getattr(Precision, n)(), fixture)
            model.pin_topk(g["detection_topk_fg"], g["detection_topk_md"])
            out = model.forward_raw(inputs(g, "detection")[:len(g.meta["sizes"])])
            e = {k: rel_err(g.like("detection_" + k, out[k].float().cpu()), g["detection_" + k]) for k in KEYS}
            print("%-18s " % n + " ".join("%-9.1e" % e[k] for k in KEYS) + "  %.1e" % max(e.values()), flush=True)
            del model
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
