"""Test-time dataset mapper: dataset dict -> the input dictionary HIPIE_IMG.forward takes.

Mirrors the `is_train == False` path of DetrDatasetMapperUni (projects/HIPIE/hipie/data/coco_dataset_mapper_uni.py:174-311 set-up,
:453-600 __call__) without detectron2's data stack:
  * read the image (detectron2.data.detection_utils.read_image: PIL, EXIF orientation, cfg.INPUT.FORMAT), check its size against
    the dataset dict, ResizeShortestEdge(MIN_SIZE_TEST, MAX_SIZE_TEST, "choice") (build_transform_gen :133-157; arithmetic and the
    PIL bilinear filter in hipie_amd/predictor.py), HWC -> CHW tensor (transform_img :312-335);
  * task "detection" with LANG_GUIDE_DET (:561-566): `expressions` = the dataset's class caption, `positive_map_label_to_token`
    (create_queries_and_maps, hipie_amd/prompts.py), `is_thing` ({class index: thing?}, index 0 = background -> False;
    register_segm_dataset :187-215, register_seginw_dataset :217-235: every class a thing) and `open_seg_labels` (the label file's
    [{"id", "name"}], what MaskCLIP prompts from);
  * any other task: `is_thing` = always true (:584-585), expressions as given.
Vocabularies are registered by name from category lists, or from the reference installation's openseg_labels files when they are
present (hipie_amd.open_vocab.load_openseg_labels).  Host-side code; nothing here touches the device.
"""
import copy

import numpy as np
import torch

from . import prompts
from .predictor import resize_image, resize_shortest_edge_shape


def read_image(file_name, fmt="RGB"):
    """detectron2.data.detection_utils.read_image: PIL, EXIF transpose, convert to `fmt` ("RGB" | "BGR" | "L")."""
    from PIL import Image, ImageOps
    with open(file_name, "rb") as f:
        img = ImageOps.exif_transpose(Image.open(f))
        if fmt == "L":
            return np.asarray(img.convert("L"))[:, :, None]
        arr = np.asarray(img.convert("RGB"))
    return arr[:, :, ::-1] if fmt == "BGR" else arr


def cat2ind_panoptics(categories):
    """coco_dataset_mapper_uni.py:42-50: {0: "__background__", 1: first name, ...} without the "invalid_class_id" entries."""
    out, i = {0: "__background__"}, 1
    for c in categories:
        if c["name"] != "invalid_class_id":
            out[i] = c["name"]
            i += 1
    return out


class TestTimeMapper(object):
    __test__ = False              # not a pytest class

    def __init__(self, tokenizer, min_size_test=800, max_size_test=1333, img_format="RGB", lang_guide_det=True):
        self.tokenizer = tokenizer
        self.min_size_test, self.max_size_test, self.img_format = min_size_test, max_size_test, img_format
        self.lang_guide_det = lang_guide_det
        self.ind_to_class_dict, self.is_thing, self.prompt_test_dict = {}, {}, {}
        self.positive_map_label_to_token_dict, self.open_seg_labels = {}, {}
        self.always_true = {k: True for k in range(200)}                      # :311

    @classmethod
    def from_yacs(cls, cfg, tokenizer):
        return cls(tokenizer, cfg.INPUT.MIN_SIZE_TEST, cfg.INPUT.MAX_SIZE_TEST, cfg.INPUT.FORMAT, cfg.MODEL.LANG_GUIDE_DET)

    def register_dataset(self, name, open_seg_labels, thing_class_ids=None):
        """open_seg_labels: [{"id", "name"}] (get_openseg_labels).  thing_class_ids: contiguous ids (0-based) of the thing classes
        as in MetadataCatalog.thing_dataset_id_to_contiguous_id.values() (register_segm_dataset); None = every class is a thing
        (register_seginw_dataset)."""
        labels = [c for c in open_seg_labels if c["name"] != "invalid_class_id"]
        i2c = cat2ind_panoptics(labels)
        self.ind_to_class_dict[name] = i2c
        if thing_class_ids is None:
            self.is_thing[name] = {k: True for k in range(len(labels) + 1)}
        else:
            things = set(int(t) for t in thing_class_ids)
            self.is_thing[name] = {k: (k - 1 in things) for k in i2c}
            self.is_thing[name][0] = False
        self.open_seg_labels[name] = labels
        self.prompt_test_dict[name], self.positive_map_label_to_token_dict[name] = prompts.create_queries_and_maps(labels, self.tokenizer)
        return self

    def register_from_label_file(self, name, thing_class_ids=None, roots=None):
        from .open_vocab import load_openseg_labels
        labels = load_openseg_labels(name, prompt_engineered=False, roots=roots)
        if labels is None:
            raise FileNotFoundError("openseg_labels/%s.txt not found (set HIPIE_ASSETS to the reference's label directory)" % name)
        return self.register_dataset(name, labels, thing_class_ids)

    def transform_img(self, image):
        """ResizeShortestEdge + HWC -> CHW (transform_img :312-335 with crop_gen None): returns (tensor, (h, w))."""
        h, w = image.shape[:2]
        nh, nw = resize_shortest_edge_shape(h, w, self.min_size_test, self.max_size_test)
        image = resize_image(np.ascontiguousarray(image), nh, nw)
        return torch.as_tensor(np.ascontiguousarray(image.transpose(2, 0, 1))), (nh, nw)

    def __call__(self, dataset_dict):
        d = copy.deepcopy(dataset_dict)
        image = d.pop("image_array", None)
        if image is None:
            image = read_image(d["file_name"], self.img_format)
        if "width" in d or "height" in d:                                      # utils.check_image_size
            if (d.get("height", image.shape[0]), d.get("width", image.shape[1])) != image.shape[:2]:
                raise ValueError("Mismatched image shape for %s: got %s, expect %s" % (
                    d.get("file_name", "<array>"), image.shape[:2], (d.get("height"), d.get("width"))))
        d.setdefault("height", image.shape[0])
        d.setdefault("width", image.shape[1])
        d["image"], _ = self.transform_img(image)
        d.pop("annotations", None)
        d.pop("sem_seg_file_name", None)
        task = d.get("task")
        if self.lang_guide_det and task == "detection":
            name = d["dataset_name"]
            if name not in self.prompt_test_dict:
                raise KeyError("dataset %r is not registered with this mapper" % name)
            d["expressions"] = self.prompt_test_dict[name]
            d["is_thing"] = self.is_thing[name]
            d["open_seg_labels"] = self.open_seg_labels[name]
            d["positive_map_label_to_token"] = self.positive_map_label_to_token_dict[name]
        else:
            d["is_thing"] = self.always_true
        return d
