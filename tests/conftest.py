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


def G(seed):
    """A fresh CPU generator: every random draw of the suite names its own seed (``torch.rand(..., generator=G(n)).to(dev)``), so no test
    depends on the process-wide random state or on which tests ran before it.  (Round 4 pinned a 1-in-25 GPU failure of
    test_graphed_train_step_equals_eager with a blanket autouse ``torch.manual_seed``; round 5 derived that test's tolerances from a
    200-seed sweep instead -- tools/graphed_step_seed_sweep.py, profiles/round5_graphed_step_seed_sweep.txt -- and removed the blanket.)"""
    return torch.Generator().manual_seed(int(seed))


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
