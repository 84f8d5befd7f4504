"""Single-image entry point: the input recipe of the reference's demo predictor, HIPIEPredictor.__call__
(projects/HIPIE/predictor.py:324-371), in front of HIPIE_IMG.forward:

    BGR -> RGB (cfg.INPUT.FORMAT "RGB"), ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST), HWC uint8 -> CHW float32,
    task "detection": caption + positive_map_label_to_token from the category names (hipie_amd/prompts.py),
    task "grounding": the referring expression as is, is_thing {1: True},
    predictions = model([inputs])[0].

The resize is detectron2's: ResizeShortestEdge.get_output_shape (detectron2/data/transforms/augmentation_impl.py:175-195) and
ResizeTransform.apply_image for uint8 images = PIL bilinear (detectron2/data/transforms/transform.py:112-125).  Host-side work;
the tensor handed to the model is moved to the device by HIPIE_IMG.preprocess_image.
"""
import numpy as np
import torch

from . import prompts


def resize_shortest_edge_shape(oldh, oldw, short_edge_length, max_size):
    """(new_h, new_w) of ResizeShortestEdge: shortest edge -> short_edge_length unless the longest would exceed max_size."""
    h, w = oldh, oldw
    size = short_edge_length * 1.0
    scale = size / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh, neww = newh * scale, neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def resize_image(img, newh, neww):
    """uint8 (H,W,3) -> (newh,neww,3) with PIL's bilinear filter (what detectron2's ResizeTransform does for uint8 input)."""
    from PIL import Image
    if img.dtype != np.uint8:
        raise TypeError("resize_image expects a uint8 image (the reference's demo path); got %s" % img.dtype)
    return np.asarray(Image.fromarray(np.ascontiguousarray(img)).resize((neww, newh), Image.BILINEAR))


class HIPIEPredictor(object):
    """model: a HIPIE_IMG (or anything with the same forward(list[dict]) contract).  tokenizer: the BERT tokenizer used to
    build class prompts (defaults to model.tokenizer).  categories: default vocabulary for task "detection" as
    [{"name": str, "isthing": 0|1}], e.g. COCO panoptic's 133 entries."""

    def __init__(self, model, tokenizer=None, min_size_test=800, max_size_test=1333, input_format="RGB", categories=None):
        self.model = model
        self.tokenizer = tokenizer if tokenizer is not None else getattr(model, "tokenizer", None)
        self.min_size_test, self.max_size_test = min_size_test, max_size_test
        if input_format not in ("RGB", "BGR"):
            raise ValueError("input_format must be RGB or BGR")
        self.input_format = input_format
        self.categories = categories
        self._prompt_cache = {}

    @classmethod
    def from_yacs(cls, model, cfg, **kw):
        return cls(model, min_size_test=cfg.INPUT.MIN_SIZE_TEST, max_size_test=cfg.INPUT.MAX_SIZE_TEST,
                   input_format=cfg.INPUT.FORMAT, **kw)

    def _prompt(self, categories):
        key = tuple((c["name"], c.get("isthing", 1)) for c in categories)
        if key not in self._prompt_cache:
            if self.tokenizer is None:
                raise RuntimeError("HIPIEPredictor: a tokenizer is needed to build class prompts")
            self._prompt_cache[key] = prompts.create_queries_and_maps(categories, self.tokenizer)
        return self._prompt_cache[key]

    def build_inputs(self, original_image, task, expressions=None, test_categories=None, test_is_thing=None, open_seg_labels=None):
        """original_image: (H,W,3) uint8 in BGR order (cv2.imread).  Returns the one-element batched_inputs list."""
        if original_image.ndim != 3 or original_image.shape[2] != 3:
            raise ValueError("original_image must be (H, W, 3)")
        if self.input_format == "RGB":
            original_image = original_image[:, :, ::-1]
        height, width = original_image.shape[:2]
        newh, neww = resize_shortest_edge_shape(height, width, self.min_size_test, self.max_size_test)
        image = resize_image(original_image, newh, neww)
        image = torch.as_tensor(image.astype("float32").transpose(2, 0, 1))
        if task == "detection":
            cats = test_categories if test_categories is not None else self.categories
            if cats is None:
                raise ValueError("task 'detection' needs test_categories (or a default vocabulary at construction)")
            caption, pmap = self._prompt(cats)
            is_thing = test_is_thing if test_is_thing is not None else {i + 1: bool(c.get("isthing", 1)) for i, c in enumerate(cats)}
            inputs = {"image": image, "height": height, "width": width, "task": task, "expressions": caption, "is_thing": is_thing,
                      "positive_map_label_to_token": pmap, "open_seg_labels": open_seg_labels}
        elif task == "grounding":
            if expressions is None:
                raise ValueError("task 'grounding' needs an expression")
            inputs = {"image": image, "height": height, "width": width, "task": task, "expressions": expressions, "is_thing": {1: True}}
        else:
            raise ValueError("Unsupported task. task must be in [\"detection\", \"grounding\"]")
        return [inputs]

    @torch.no_grad()
    def __call__(self, original_image, task, expressions=None, test_categories=None, open_seg_labels=None, test_is_thing=None):
        return self.model(self.build_inputs(original_image, task, expressions, test_categories, test_is_thing, open_seg_labels))[0]
