"""prints the measured error of every side-policy e2e comparison next to its bound (how much room the GPU suite's bounds have on this box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
torch.set_grad_enabled(False)
import test_gpu_e2e as T
from hipie_amd.config import Precision
from util import rel_err
for fixture, tasks, pols in (("e2e_tiny", ("detection", "grounding"), (("parity", 1e-3), ("split3", 1e-3), ("fast", 8e-3))),
                             ("e2e_r50_tiny", ("detection", "grounding"), (("split3", 1e-3), ("parity", 1e-3), ("fast", 8e-3), ("bf16", 8e-2))),
                             ("e2e_r50_512", ("detection", "grounding"), (("split3", 1e-3), ("parity", 2e-3)))):
    for pol, tol in pols:
        g, model = T.build(getattr(Precision, pol)(), fixture)
        for task in tasks:
            model.pin_topk(g[task + "_topk_fg"], g[task + "_topk_md"])
            out = model.forward_raw(T.inputs(g, task))
            errs = {k: rel_err(g.like(task + "_" + k, out[k].float().cpu()), g[task + "_" + k]) for k in T.KEYS}
            k = max(errs, key=errs.get)
            print("%-13s %-10s %-7s worst %-22s %.2e  bound %.0e  (%.0f %% of the bound)" % (fixture, task, pol, k, errs[k], tol, 100 * errs[k] / tol), flush=True)
