"""tests/golden/eval_cfgs.json: the 11 shipped eval configurations of the reference, merged by the reference's OWN config code
(detectron2 defaults -> add_hipie_config -> yaml with _BASE_ inheritance; the MaskDINO sub-configuration as build_maskdino assembles it)
through tests/golden/cfg_shim.py, flattened to "A.B.C": value tables.  Run in the build container (needs /root/reference):

    python tests/golden/gen_cfg_golden.py
"""
import json
import os

import cfg_shim

HERE = os.path.dirname(os.path.abspath(__file__))
KEEP = ("MODEL.", "TEST.", "INPUT.", "DATASETS.TEST", "VERSION")


def tables():
    out = {"eval": {}, "maskdino": {}}
    for y in cfg_shim.eval_yamls():
        cfg = cfg_shim.hipie_cfg(y)
        flat = cfg_shim.flatten(cfg)
        out["eval"][os.path.basename(y)] = {k: v for k, v in sorted(flat.items()) if k.startswith(KEEP)}
        mp = cfg.MODEL.MASKDINO.CONFIG_PATH
        if mp not in out["maskdino"]:
            md = cfg_shim.flatten(cfg_shim.maskdino_cfg(mp))
            out["maskdino"][mp] = {k: v for k, v in sorted(md.items()) if k.startswith(("MODEL.MaskDINO.", "MODEL.SEM_SEG_HEAD."))}
    return out


if __name__ == "__main__":
    t = tables()
    path = os.path.join(HERE, "eval_cfgs.json")
    with open(path, "w") as f:
        json.dump(t, f, indent=0, sort_keys=True)
    print("wrote %s: %d eval configurations, %d MaskDINO configuration(s), %.0f KB"
          % (path, len(t["eval"]), len(t["maskdino"]), os.path.getsize(path) / 1024))
