import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load_params(module, g, prefix="p."):
    """Load the exported reference parameters/buffers (`p.<state_dict key>`) into `module`."""
    import torch
    sd = {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in g.items() if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert not [m for m in missing if "unk_mask" not in m], missing
    return module


@pytest.fixture(scope="session")
def golden():
    return load_golden
