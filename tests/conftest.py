import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("DEMO_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (only in the build container)")


def has_gpu():
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    skip_gpu = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords and not has_gpu():
            item.add_marker(skip_gpu)


def load_golden(name):
    """-> nested dict: top-level arrays and 'group::key' entries regrouped as dicts of torch tensors."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = torch.from_numpy(np.asarray(z[k]))
        if "::" in k:
            g, kk = k.split("::", 1)
            out.setdefault(g, {})[kk] = v
        else:
            out[k] = v
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
