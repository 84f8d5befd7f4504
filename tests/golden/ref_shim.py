"""Import shim used ONLY by tests/golden/gen_golden.py, in the authoring container.

It lets the reference's hot-path *modules* (not the whole product) be imported from
/root/reference on a box that has neither detectron2's dependencies (fvcore, yacs,
iopath), nor torchvision / timm / fairscale, nor a GPU.  Nothing here ships to the GPU
box and nothing under hipie_amd/ or oracle/ imports it.

What it does:
  * registers a namespace package ``hipie_ref`` whose __path__ is the reference's
    ``projects/HIPIE/hipie`` directory, so ``hipie/__init__.py`` (which pulls in the data
    pipeline) is skipped;
  * installs a meta-path finder that fabricates empty stand-in modules for the missing
    third-party packages; a handful of names that the hot path really uses get small
    functional stand-ins (Conv2d with norm/activation, ShapeSpec, registries, DropPath,
    Mlp, c2_xavier_fill ...);
  * routes both copies of ``MSDeformAttnFunction`` to the reference's own pure-PyTorch
    formulation ``ms_deform_attn_core_pytorch`` (the CUDA extension cannot be built here).

The stand-ins only replace *third-party* code; every line of arithmetic that ends up in a
golden vector is executed from the reference's own files.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys
import types
from collections import namedtuple

import torch
import torch.nn as nn
import torch.nn.functional as F

REF_ROOT = "/root/reference"
REF_HIPIE = REF_ROOT + "/projects/HIPIE/hipie"

_STUB_ROOTS = (
    "detectron2", "fvcore", "torchvision", "timm", "fairscale", "cv2", "skimage", "pycocotools",
    "panopticapi", "lvis", "wandb", "iopath", "yacs", "omegaconf", "shapely", "segment_anything",
    "open_clip", "MultiScaleDeformableAttention", "termcolor", "tabulate_stub",
)


class _Anything:
    """Inert placeholder: callable, subscriptable, usable as decorator; never used for math."""

    def __init__(self, name="?"):
        self._name = name

    def __call__(self, *a, **k):
        if len(a) == 1 and not k and (isinstance(a[0], type) or callable(a[0])):
            return a[0]  # decorator use
        return _Anything(self._name + "()")

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _Anything(self._name + "." + item)

    def __getitem__(self, item):
        return _Anything(self._name + "[]")

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        full = self.__name__ + "." + item
        if full in sys.modules:
            return sys.modules[full]
        val = _Anything(full)
        setattr(self, item, val)
        return val


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


# ---------------------------------------------------------------------------- stand-ins
ShapeSpec = namedtuple("ShapeSpec", ["channels", "height", "width", "stride"], defaults=(None,) * 4)


class _Registry:
    def __init__(self, name="r"):
        self._d = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._d[o.__name__] = o
                return o
            return deco
        self._d[obj.__name__] = obj
        return obj

    def get(self, name):
        return self._d[name]


class _Mlp(nn.Module):
    """timm.models.layers.Mlp (fc1 -> act -> fc2; dropout 0)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _DropPath(nn.Identity):
    def __init__(self, *a, **k):
        super().__init__()


def _c2_xavier_fill(m):
    nn.init.kaiming_uniform_(m.weight, a=1)
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def _c2_msra_fill(m):
    nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    if m.bias is not None:
        nn.init.constant_(m.bias, 0)


def _box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


class _CNNBlockBase(nn.Module):
    def __init__(self, in_channels, out_channels, stride):
        super().__init__()
        self.in_channels, self.out_channels, self.stride = in_channels, out_channels, stride


_REF_LAYERS = None


def ref_d2_layers():
    """the reference's own vendored detectron2/layers/wrappers.py (Conv2d with norm / activation, ConvTranspose2d) and
    detectron2/layers/batch_norm.py (FrozenBatchNorm2d, get_norm, LayerNorm) loaded from their files; only their imports of
    packages that are absent here (fvcore.nn.distributed, detectron2.utils.comm/env: used by the SyncBN variants) are stubs."""
    global _REF_LAYERS
    if _REF_LAYERS is None:
        import importlib.util
        pkg = types.ModuleType("ref_d2layers")
        pkg.__path__ = []
        sys.modules["ref_d2layers"] = pkg
        out = {}
        for name in ("wrappers", "batch_norm"):
            spec = importlib.util.spec_from_file_location("ref_d2layers." + name, REF_ROOT + "/detectron2/layers/%s.py" % name)
            mod = importlib.util.module_from_spec(spec)
            sys.modules["ref_d2layers." + name] = mod
            spec.loader.exec_module(mod)
            out[name] = mod
        _REF_LAYERS = out
    return _REF_LAYERS


def _populate(m):
    n = m.__name__
    if n == "detectron2.layers":
        ref = ref_d2_layers()
        m.Conv2d = ref["wrappers"].Conv2d
        m.ConvTranspose2d = ref["wrappers"].ConvTranspose2d
        m.ShapeSpec = ShapeSpec
        m.get_norm = ref["batch_norm"].get_norm
        m.FrozenBatchNorm2d = ref["batch_norm"].FrozenBatchNorm2d
        m.CNNBlockBase = _CNNBlockBase
    elif n == "detectron2.layers.batch_norm":
        m.get_norm = ref_d2_layers()["batch_norm"].get_norm
    elif n == "fvcore.nn.distributed":
        m.differentiable_all_reduce = None               # SyncBN only
    elif n == "detectron2.utils.env":
        m.TORCH_VERSION = tuple(int(x) for x in torch.__version__.split(".")[:2])
    elif n == "detectron2.utils":
        m.comm = importlib.import_module("detectron2.utils.comm")
        m.env = importlib.import_module("detectron2.utils.env")
    elif n == "detectron2.modeling":
        m.BACKBONE_REGISTRY = _Registry()
        m.SEM_SEG_HEADS_REGISTRY = _Registry()
        m.META_ARCH_REGISTRY = _Registry()
        m.Backbone = nn.Module
        m.ShapeSpec = ShapeSpec
    elif n == "detectron2.modeling.backbone.fpn":
        m._assert_strides_are_log2_contiguous = lambda s: None
    elif n == "detectron2.config":
        m.configurable = lambda f=None, **k: f
    elif n == "detectron2.utils.registry":
        m.Registry = _Registry
    elif n == "fvcore.nn.weight_init" or n == "fvcore.nn":
        m.c2_xavier_fill = _c2_xavier_fill
        m.c2_msra_fill = _c2_msra_fill
        if n == "fvcore.nn":
            m.weight_init = importlib.import_module("fvcore.nn.weight_init")
    elif n == "timm.models.layers":
        m.DropPath = _DropPath
        m.Mlp = _Mlp
    elif n == "torchvision.ops.boxes":
        m.box_area = _box_area
    elif n == "torchvision":
        m.__version__ = "0.99.0"
    elif n == "torchvision.transforms":
        m.Compose = _TVCompose
    elif n == "open_clip":
        m.create_model_and_transforms = _oc_create_model_and_transforms
        m.tokenize = _oc_tokenize
    elif n == "detectron2.utils.comm":
        m.get_local_rank = lambda: 0
        m.synchronize = lambda: None


# ---------------------------------------------------------------------------- open_clip stand-in (MaskCLIP golden)
# open-clip-torch is THIRD-PARTY (pinned to 2.0.2 by ODISE, the origin of hipie/open_vocab/clip.py), absent from /root/reference and
# from this image.  The classes below follow its published architecture (open_clip/model.py, v2.0.2: CLIP, VisualTransformer,
# Transformer, ResidualAttentionBlock around torch.nn.MultiheadAttention, QuickGELU for the OpenAI weights) closely enough for the
# reference's MaskCLIP / ClipAdapter to run on top: same attribute and parameter names, attention executed by torch's own
# nn.MultiheadAttention.  Size and vocabulary come from OPEN_CLIP_CFG (a tiny configuration for the fixtures).
OPEN_CLIP_CFG = dict(width=128, layers=2, heads=2, patch=14, image_size=84, embed_dim=32, text_width=64, text_layers=2, text_heads=2,
                     context=16, vocab=512, quick_gelu=True)


class _QuickGELU(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(1.702 * x)


class _OCBlock(nn.Module):
    def __init__(self, d, heads, quick_gelu):
        super().__init__()
        from collections import OrderedDict
        self.ln_1 = nn.LayerNorm(d)
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d, d * 4)), ("gelu", _QuickGELU() if quick_gelu else nn.GELU()),
                                              ("c_proj", nn.Linear(d * 4, d))]))

    def forward(self, x, attn_mask=None):
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _OCTransformer(nn.Module):
    def __init__(self, d, layers, heads, quick_gelu):
        super().__init__()
        self.resblocks = nn.ModuleList([_OCBlock(d, heads, quick_gelu) for _ in range(layers)])

    def forward(self, x, attn_mask=None):
        for r in self.resblocks:
            x = r(x, attn_mask=attn_mask)
        return x


class _OCVisual(nn.Module):
    def __init__(self, c):
        super().__init__()
        d, g = c["width"], c["image_size"] // c["patch"]
        self.image_size = c["image_size"]
        self.conv1 = nn.Conv2d(3, d, kernel_size=c["patch"], stride=c["patch"], bias=False)
        self.class_embedding = nn.Parameter(d ** -0.5 * torch.randn(d))
        self.positional_embedding = nn.Parameter(d ** -0.5 * torch.randn(g * g + 1, d))
        self.ln_pre = nn.LayerNorm(d)
        self.transformer = _OCTransformer(d, c["layers"], c["heads"], c["quick_gelu"])
        self.ln_post = nn.LayerNorm(d)
        self.proj = nn.Parameter(d ** -0.5 * torch.randn(d, c["embed_dim"]))


class _OCCLIP(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.context_length = c["context"]
        self.visual = _OCVisual(c)
        self.transformer = _OCTransformer(c["text_width"], c["text_layers"], c["text_heads"], c["quick_gelu"])
        self.token_embedding = nn.Embedding(c["vocab"], c["text_width"])
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(c["context"], c["text_width"]))
        self.ln_final = nn.LayerNorm(c["text_width"])
        self.text_projection = nn.Parameter(c["text_width"] ** -0.5 * torch.randn(c["text_width"], c["embed_dim"]))
        self.logit_scale = nn.Parameter(torch.ones([]) * 2.659)
        mask = torch.empty(c["context"], c["context"]).fill_(float("-inf")).triu_(1)
        self.register_buffer("attn_mask", mask, persistent=False)

    def encode_text(self, text):
        x = self.token_embedding(text) + self.positional_embedding
        x = self.transformer(x.permute(1, 0, 2), attn_mask=self.attn_mask).permute(1, 0, 2)
        x = self.ln_final(x)
        return x[torch.arange(x.shape[0]), text.argmax(dim=-1)] @ self.text_projection


class _TVNormalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean).view(1, 3, 1, 1), torch.tensor(std).view(1, 3, 1, 1)

    def __call__(self, x):
        return (x - self.mean) / self.std


class _TVSameSize:
    """Resize / CenterCrop of the OpenAI preprocess: MaskCLIP only calls them on images it has already resized to the model size."""

    def __init__(self, size):
        self.size = size

    def __call__(self, x):
        assert tuple(x.shape[-2:]) == (self.size, self.size), (tuple(x.shape), self.size)
        return x


class _TVCompose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def _oc_create_model_and_transforms(model_name=None, pretrained=None, device=None, **kw):
    c = OPEN_CLIP_CFG
    model = _OCCLIP(c).eval()
    pre = _TVCompose([_TVSameSize(c["image_size"]), _TVSameSize(c["image_size"]), (lambda x: x), (lambda x: x),
                      _TVNormalize((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711))])
    return model, None, pre


def _oc_tokenize(texts):
    import _synth
    return _synth.clip_tokenize(texts, OPEN_CLIP_CFG["context"], OPEN_CLIP_CFG["vocab"])


def ref_clip():
    """the reference's hipie/open_vocab/clip.py itself (ClipAdapter, MaskCLIP, build_clip_text_embed) over the stand-ins above."""
    install()
    return ref("open_vocab.clip")


_INSTALLED = False


def install():
    global _INSTALLED
    if _INSTALLED:
        return
    _INSTALLED = True
    # transformers probes torchvision on import: load what the reference needs first.
    import transformers  # noqa: F401
    import transformers.models.bert.modeling_bert  # noqa: F401
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for name in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(pu, "find_pruneable_heads_and_indices", lambda *a, **k: None)
    sys.meta_path.insert(0, _StubFinder())
    pkg = types.ModuleType("hipie_ref")
    pkg.__path__ = [REF_HIPIE]
    sys.modules["hipie_ref"] = pkg
    # both copies of the MSDA autograd function -> the reference's own pytorch formulation
    for sub in ("hipie_ref.models.deformable_detr.ops", "hipie_ref.models.maskdino.pixel_decoder.ops"):
        fn = importlib.import_module(sub + ".functions.ms_deform_attn_func")

        class _Fn:  # noqa: N801
            @staticmethod
            def apply(value, shapes, level_start, loc, attn, im2col_step, _core=fn.ms_deform_attn_core_pytorch):
                return _core(value, shapes, loc, attn)
        fn.MSDeformAttnFunction = _Fn
        mod = importlib.import_module(sub + ".modules.ms_deform_attn")
        mod.MSDeformAttnFunction = _Fn


def ref_d2_resnet():
    """load the reference's detectron2/modeling/backbone/resnet.py itself (its package imports are served by the
    stand-ins above: CNNBlockBase, Conv2d, get_norm(FrozenBN), ShapeSpec, Backbone, BACKBONE_REGISTRY, weight_init)."""
    install()
    import importlib.util
    pkg = types.ModuleType("ref_d2bb")
    pkg.__path__ = []
    sys.modules["ref_d2bb"] = pkg
    bb = types.ModuleType("ref_d2bb.backbone")
    bb.Backbone = nn.Module
    sys.modules["ref_d2bb.backbone"] = bb
    bd = types.ModuleType("ref_d2bb.build")
    bd.BACKBONE_REGISTRY = _Registry()
    sys.modules["ref_d2bb.build"] = bd
    spec = importlib.util.spec_from_file_location("ref_d2bb.resnet", REF_ROOT + "/detectron2/modeling/backbone/resnet.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_d2bb.resnet"] = mod
    spec.loader.exec_module(mod)
    return mod


def tv_batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms -- THIRD-PARTY, absent from /root/reference and from this image (the reference does not
    pin a torchvision version; INSTALL follows detectron2's "torchvision matching the PyTorch install").  Restated from the
    published algorithm: boxes.numel() <= 4000 -> coordinate trick (offset every box by idx * (max_coordinate + 1)), then
    greedy NMS (torchvision/csrc/ops/cpu/nms_kernel.cpp: stable descending sort by score; IoU = inter / (a_i + a_j - inter),
    suppress when IoU > threshold); keep indices are returned in decreasing-score order."""
    if boxes.numel() == 0:
        return torch.empty((0,), dtype=torch.int64)
    if boxes.numel() > 4000:
        keep_mask = torch.zeros_like(scores, dtype=torch.bool)
        for c in torch.unique(idxs):
            ci = torch.where(idxs == c)[0]
            keep_mask[ci[_tv_nms(boxes[ci], scores[ci], iou_threshold)]] = True
        k = torch.where(keep_mask)[0]
        return k[scores[k].sort(descending=True)[1]]
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    return _tv_nms(boxes + offsets[:, None], scores, iou_threshold)


def _tv_nms(boxes, scores, thr):
    x1, y1, x2, y2 = boxes.unbind(1)
    areas = (x2 - x1) * (y2 - y1)
    order = scores.sort(dim=0, descending=True, stable=True)[1]
    n = boxes.shape[0]
    dead = [False] * n
    keep = []
    for _i in range(n):
        i = int(order[_i])
        if dead[i]:
            continue
        keep.append(i)
        for _j in range(_i + 1, n):
            j = int(order[_j])
            if dead[j]:
                continue
            w = torch.clamp(torch.min(x2[i], x2[j]) - torch.max(x1[i], x1[j]), min=0)
            h = torch.clamp(torch.min(y2[i], y2[j]) - torch.max(y1[i], y1[j]), min=0)
            inter = w * h
            if float(inter / (areas[i] + areas[j] - inter)) > thr:
                dead[j] = True
    return torch.tensor(keep, dtype=torch.int64)


def ref_hipie_img():
    """the reference's hipie_img.py itself (for HIPIE_IMG.inference / panoptic_inference / semantic_inference /
    convert_grounding_to_od_logits and ddetrs.segmentation_postprocess), with the data pipeline / SAM / MaskCLIP
    sub-packages it imports at module level left as inert stubs, detectron2's own (vendored, dependency-free)
    structures/{boxes,instances}.py loaded from /root/reference, and torchvision's batched_nms restated above."""
    install()
    import importlib.util
    for n in ("hipie_ref.data", "hipie_ref.data.coco_dataset_mapper_uni", "hipie_ref.models.sam", "hipie_ref.open_vocab.clip"):
        if n not in sys.modules:
            mod = _StubModule(n)
            mod.__path__ = []
            sys.modules[n] = mod
    m = ref("hipie_img")
    st = {}
    for name in ("boxes", "instances"):
        spec = importlib.util.spec_from_file_location("ref_d2st." + name, REF_ROOT + "/detectron2/structures/%s.py" % name)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        st[name] = mod
    dd = ref("models.ddetrs")
    for tgt in (m, dd):
        tgt.Instances = st["instances"].Instances
        tgt.Boxes = st["boxes"].Boxes
    m.retry_if_cuda_oom = lambda f: f
    m.ops = types.SimpleNamespace(batched_nms=tv_batched_nms)
    return m


def ref(modname):
    """import ``hipie_ref.<modname>`` (a module path below projects/HIPIE/hipie)."""
    install()
    return importlib.import_module("hipie_ref." + modname)
