import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz -> nested dict of torch tensors ('a::b' keys become d['a']['b'])."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    out = {}
    for k in z.files:
        v = torch.from_numpy(np.asarray(z[k]))
        if '::' in k:
            a, b = k.split('::', 1)
            out.setdefault(a, {})[b] = v
        else:
            out[k] = v
    return out


@pytest.fixture(scope='session')
def golden():
    return load_golden
