"""CPU: the post-processing oracle (oracle/post.py) against the golden produced by the reference's own
HIPIE_IMG.inference / panoptic_inference / segmentation_postprocess (tests/golden/gen_golden.py post)."""
import numpy as np
import pytest
import torch

import _synth
from oracle import post as op
from util import Golden, rel_err


def post_case(g, cname):
    P, c = g.meta["post"], g.meta["cases"][cname]
    sizes = [tuple(s) for s in P["sizes"]]
    a22 = _synth.synth_a22(sizes, P["n_bg"], P["n_fg"], P["n_md"], P["L"], seed=P["seed"])
    pmap = {1: [0]} if c["task"] == "grounding" else {int(k): v for k, v in g.meta["pmap"].items()}
    is_thing = {int(k): v for k, v in P["is_thing"].items()}
    out_hw = [tuple(x) for x in c["out_hw"]] if c["out_hw"] else sizes
    kw = dict(use_bg_for_pano=c["use_bg_for_pano"], bg_cls_agnostic=c["bg_cls_agnostic"], max_pool=c["max_pool"])
    return a22, sizes, pmap, is_thing, out_hw, kw, c["task"], P["n_bg"]


def check_against_golden(g, cname, res, tol=1e-5, mask_mismatch=0):
    for i, r in enumerate(res):
        pre = "%s_%d_" % (cname, i)
        inst = r["instances"]
        assert torch.equal(inst["classes"].cpu().long(), g[pre + "classes"])
        assert rel_err(inst["scores"].cpu(), g[pre + "scores"]) < tol
        assert rel_err(inst["boxes"].cpu(), g[pre + "boxes"]) < tol
        shape = g.meta[pre + "masks_shape"]
        want = np.unpackbits(g[pre + "masks"].numpy())[:int(np.prod(shape))].reshape(shape).astype(bool)
        got = inst["masks"].cpu().numpy().astype(bool)
        assert got.shape == want.shape
        assert (got != want).sum() <= mask_mismatch * want.size
        if (pre + "panoptic") in g.keys():
            pan, info = r["panoptic_seg"]
            assert info == g.meta[pre + "segments"]
            assert (pan.cpu().long() != g[pre + "panoptic"].long()).sum() <= mask_mismatch * pan.numel()
            assert rel_err(g.like(pre + "semseg", r["sem_seg"].cpu().float()), g[pre + "semseg"]) < max(tol, 1e-5)


@pytest.mark.parametrize("cname", ["default", "evalyaml", "grounding"])
def test_post_oracle_matches_reference(cname):
    g = Golden("post")
    a22, sizes, pmap, is_thing, out_hw, kw, task, nbg = post_case(g, cname)
    res = op.inference(a22, sizes, pmap, task, [is_thing] * len(sizes), out_sizes=out_hw, num_bg=nbg, **kw)
    check_against_golden(g, cname, res)
