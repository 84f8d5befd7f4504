"""shared helpers for the tests (fixtures loading, error metrics)."""
import json
import os

import numpy as np
import torch

import _synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(bytes(self.z["meta"]).decode())

    def __getitem__(self, k):
        return torch.from_numpy(self.z[k])

    def keys(self):
        return [k for k in self.z.files if k != "meta"]

    def like(self, key, t):
        """bring a freshly computed full tensor to the fixture's (possibly subsampled) form."""
        sub = self.meta.get("subsampled", {})
        if key in sub:
            step, shape = sub[key]
            assert list(t.shape) == shape, (key, list(t.shape), shape)
            return _synth.subsample(t.contiguous(), step)
        return t


def rel_err(a, b):
    """max|a-b| / max|b| : the 'rel' of BASELINE.json's 1e-3 tolerance (error relative to the output scale)."""
    a, b = a.double(), b.double()
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin)
    if not fin.all():
        assert torch.equal(a[~fin], b[~fin])
        a, b = a[fin], b[fin]
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
