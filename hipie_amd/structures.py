"""Result containers of the meta-arch's output (detectron2.structures.Instances / Boxes, D2/structures/instances.py:9-196,
boxes.py:130-310).  When detectron2 is importable its own classes are used, so evaluators and visualisers receive exactly
the types they expect; otherwise (the GPU box has no detectron2) these minimal look-alikes carry the same
fields with the same accessors (``.tensor``, ``.has``, ``.get_fields``, ``len``, ``.to``, attribute access)."""
import torch

try:                                                      # pragma: no cover - exercised only where detectron2 exists
    from detectron2.structures import Boxes, Instances    # noqa: F401
except Exception:                                         # noqa: BLE001 - any import problem means "not available"
    class Boxes:
        def __init__(self, tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
            if tensor.numel() == 0:
                tensor = tensor.reshape((-1, 4))
            assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
            self.tensor = tensor

        def __len__(self):
            return self.tensor.shape[0]

        def to(self, *a, **k):
            return Boxes(self.tensor.to(*a, **k))

        def __getitem__(self, item):
            if isinstance(item, int):
                return Boxes(self.tensor[item].view(1, -1))
            return Boxes(self.tensor[item])

        def area(self):
            b = self.tensor
            return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

        def __repr__(self):
            return "Boxes(%s)" % self.tensor

    class Instances:
        def __init__(self, image_size, **fields):
            object.__setattr__(self, "_image_size", tuple(image_size))
            object.__setattr__(self, "_fields", {})
            for k, v in fields.items():
                self.set(k, v)

        @property
        def image_size(self):
            return self._image_size

        def __setattr__(self, name, val):
            if name.startswith("_"):
                object.__setattr__(self, name, val)
            else:
                self.set(name, val)

        def __getattr__(self, name):
            if name == "_fields" or name not in self._fields:
                raise AttributeError("Cannot find field '%s' in the given Instances!" % name)
            return self._fields[name]

        def set(self, name, value):
            if len(self._fields):
                assert len(self) == len(value), "Adding a field of length %d to Instances of length %d" % (len(value), len(self))
            self._fields[name] = value

        def has(self, name):
            return name in self._fields

        def get(self, name):
            return self._fields[name]

        def get_fields(self):
            return self._fields

        def to(self, *a, **k):
            r = Instances(self._image_size)
            for n, v in self._fields.items():
                r.set(n, v.to(*a, **k) if hasattr(v, "to") else v)
            return r

        def __len__(self):
            for v in self._fields.values():
                return len(v)
            raise NotImplementedError("Empty Instances does not support __len__!")

        def __getitem__(self, item):
            r = Instances(self._image_size)
            for n, v in self._fields.items():
                r.set(n, v[item])
            return r

        def __repr__(self):
            return "Instances(num_instances=%d, image_size=%s, fields=[%s])" % (
                len(self) if self._fields else 0, self._image_size, ", ".join(self._fields))
