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


@pytest.fixture(autouse=True)
def _deterministic_draws():
    """Every test starts from the same CPU and device random state: a few GPU tests draw inputs / targets on the device without a generator
    of their own, and a tolerance that holds for almost every draw (test_graphed_train_step_equals_eager compares trajectories of Adam
    steps) must not decide a run by chance."""
    torch.manual_seed(20260927)
    yield


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
