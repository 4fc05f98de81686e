import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) where no GPU exists, so a plain `pytest tests` works anywhere."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class Fixture:
    """One tests/golden/*.npz produced by tools/gen_golden.py from the real reference."""

    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.name = name
        self.cfg = json.loads(str(z["cfg"]))
        self.groups = {}
        for k in z.files:
            if k == "cfg":
                continue
            g, key = k.split("/", 1)
            self.groups.setdefault(g, {})[key] = torch.from_numpy(np.array(z[k]))

    def __getattr__(self, g):
        try:
            return self.groups[g]
        except KeyError:
            raise AttributeError(g)

    def params(self, requires_grad=True, device="cpu"):
        out = {}
        for k, v in self.groups["P"].items():
            t = v.clone().to(device)
            if requires_grad and t.is_floating_point():
                t.requires_grad_()
            out[k] = t
        return out


def golden_names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


@pytest.fixture
def load_golden():
    return Fixture
