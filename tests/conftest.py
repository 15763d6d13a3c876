import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    """tests/golden/<name>.npz -> dict of torch tensors / numpy scalars (fixtures made by make_golden.py)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        for k in z.files:
            a = z[k]
            out[k] = torch.from_numpy(a) if (a.ndim > 0 and a.dtype.kind in 'fiu') else a
    return out


def sub(d, prefix):
    """Entries of ``d`` under ``prefix`` with the prefix stripped."""
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def bn_of(params, prefix):
    return {k: params[f'{prefix}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max |a-b| / max |b|  -- the 'relative fp32' error of the north star (tensor-scale relative)."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp(min=1e-30))
