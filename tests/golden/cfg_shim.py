"""The reference's REAL configuration surface without yacs / fvcore / detectron2 installed (build container only).

What runs here is the reference's own code, imported in place from /root/reference:
  detectron2/config/defaults.py                       (the detectron2 default tree `_C`)
  projects/HIPIE/hipie/config.py::add_hipie_config      (every key the shipped yamls may set)
  projects/DeepLab/deeplab/config.py::add_deeplab_config, projects/HIPIE/hipie/models/maskdino/config.py::add_maskdino_config
over a small stand-in for `detectron2.config.CfgNode` with yacs' merge semantics (yacs/config.py: `_merge_a_into_b` -- a key that the
defaults do not define is a KeyError "Non-existent config key", values are decoded with literal_eval, tuples / lists coerce into each
other, other type changes raise) and fvcore's `_BASE_` inheritance (fvcore/common/config.py: load_yaml_with_base).  The yaml files are the
reference's shipped ones, read where they lie.  Nothing of this travels to the GPU box: tests/golden/gen_cfg_golden.py writes the merged
trees as flat key/value tables (tests/golden/eval_cfgs.json) and the GPU-side test rebuilds plain namespaces from them.
"""
import ast
import copy
import importlib.util
import os
import sys
import types

import yaml

REF = "/root/reference"
HIPIE = os.path.join(REF, "projects", "HIPIE")


class CfgNode(dict):
    """attribute-style nested dict with yacs' merge rules (a stand-in, NOT yacs)."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        for k, v in (init_dict or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def defrost(self):
        pass

    @staticmethod
    def _decode(v):
        if isinstance(v, dict):
            return CfgNode(v)
        if not isinstance(v, str):
            return v
        try:
            return ast.literal_eval(v)          # "(55100,84000)" -> tuple, as yacs' _decode_cfg_value
        except (ValueError, SyntaxError):
            return v

    @staticmethod
    def _coerce(new, old, full_key):
        if type(new) is type(old) or old is None or new is None:
            return new
        for a, b in ((list, tuple), (tuple, list)):
            if isinstance(new, a) and isinstance(old, b):
                return b(new)
        if isinstance(old, float) and isinstance(new, int) and not isinstance(new, bool):
            return float(new)                   # fvcore allows int -> float
        raise ValueError("Type mismatch (%s vs. %s) for config key: %s" % (type(old), type(new), full_key))

    def _merge(self, other, keys):
        for k, v_ in other.items():
            full = ".".join(keys + [k])
            v = self._decode(copy.deepcopy(v_))
            if k not in self:
                raise KeyError("Non-existent config key: %s" % full)
            if isinstance(self[k], CfgNode) and isinstance(v, dict):
                self[k]._merge(v, keys + [k])
            else:
                self[k] = self._coerce(v, self[k], full)

    @staticmethod
    def load_yaml_with_base(filename):
        with open(filename) as f:
            cfg = yaml.unsafe_load(f)           # detectron2's CfgNode loads unsafely by default (config.py:37-40); YAML 1.1: on / off are bools
        base = cfg.pop("_BASE_", None)
        if base is None:
            return cfg
        if not os.path.isabs(base):
            base = os.path.join(os.path.dirname(filename), base)
        out = CfgNode.load_yaml_with_base(base)

        def merge(a, b):
            for k, v in a.items():
                if isinstance(v, dict) and isinstance(b.get(k), dict):
                    merge(v, b[k])
                else:
                    b[k] = v
        merge(cfg, out)
        return out

    def merge_from_file(self, filename, allow_unsafe=True):
        self._merge(self.load_yaml_with_base(filename), [])

    def merge_from_other_cfg(self, other):
        self._merge(other, [])

    def merge_from_list(self, lst):
        for k, v in zip(lst[0::2], lst[1::2]):
            node, parts = self, k.split(".")
            for p in parts[:-1]:
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % k)
            node[parts[-1]] = self._coerce(self._decode(v), node[parts[-1]], k)


def _load(path, name, package=None):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    if package is not None:
        mod.__package__ = package
    spec.loader.exec_module(mod)
    return mod


class _Modules(object):
    """install the stand-in `detectron2.config` for the duration of the imports, then restore sys.modules"""

    NAMES = ("detectron2", "detectron2.config", "detectron2.config.config")

    def __enter__(self):
        self.saved = {n: sys.modules.get(n) for n in self.NAMES}
        d2, cfgpkg, cfgmod = types.ModuleType("detectron2"), types.ModuleType("detectron2.config"), types.ModuleType("detectron2.config.config")
        d2.__path__, cfgpkg.__path__ = [], []
        cfgmod.CfgNode = cfgpkg.CfgNode = CfgNode
        d2.config = cfgpkg
        cfgpkg.config = cfgmod
        sys.modules.update({"detectron2": d2, "detectron2.config": cfgpkg, "detectron2.config.config": cfgmod})
        return self

    def __exit__(self, *a):
        for n, m in self.saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m


def d2_defaults():
    """detectron2.config.get_cfg(): a clone of defaults.py's `_C` (detectron2/config/config.py:99-108)"""
    with _Modules():
        mod = _load(os.path.join(REF, "detectron2", "config", "defaults.py"), "detectron2.config.defaults", package="detectron2.config")
    return mod._C.clone()


def hipie_cfg(yaml_path, opts=()):
    """train_net.py:246-256 setup(): get_cfg() -> add_hipie_config -> merge_from_file -> merge_from_list"""
    with _Modules():
        add = _load(os.path.join(HIPIE, "hipie", "config.py"), "_ref_hipie_config").add_hipie_config
        cfg = d2_defaults()
        add(cfg)
    cfg.merge_from_file(yaml_path)
    cfg.merge_from_list(list(opts))
    return cfg


def maskdino_cfg(config_path):
    """hipie/models/maskdino/build.py:8-19 build_maskdino(): get_cfg() -> add_deeplab_config -> add_maskdino_config -> merge_from_file.
    config_path is cwd-relative to the reference repository root (launch.py:50-52 chdirs there)."""
    with _Modules():
        deeplab = _load(os.path.join(REF, "projects", "DeepLab", "deeplab", "config.py"), "_ref_deeplab_config").add_deeplab_config
        md = _load(os.path.join(HIPIE, "hipie", "models", "maskdino", "config.py"), "_ref_maskdino_config").add_maskdino_config
        cfg = d2_defaults()
        deeplab(cfg)
        md(cfg)
    cfg.merge_from_file(config_path if os.path.isabs(config_path) else os.path.join(REF, config_path))
    return cfg


def flatten(node, prefix=""):
    out = {}
    for k, v in node.items():
        if isinstance(v, dict):
            out.update(flatten(v, prefix + k + "."))
        else:
            out[prefix + k] = list(v) if isinstance(v, tuple) else v
    return out


def eval_yamls():
    d = os.path.join(HIPIE, "configs", "eval")
    return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".yaml"))
