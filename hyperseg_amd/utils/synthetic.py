"""Deterministic, name-keyed weights: every tensor is a function of (its state-dict key, its shape, seed),
so two independently written models with the same parameter names get identical weights without any
checkpoint travelling (used to pin the stock-PyTorch encoder / context head to the reference)."""
import hashlib
import math

import torch


def tensor_for(key, shape, seed=0):
    h = int.from_bytes(hashlib.sha256(f'{seed}:{key}'.encode()).digest()[:8], 'little') % (2 ** 63)
    g = torch.Generator().manual_seed(h)
    if key.endswith('running_var'):
        return torch.rand(shape, generator=g) * 0.4 + 0.8
    if key.endswith('running_mean'):
        return torch.randn(shape, generator=g) * 0.1
    if len(shape) == 1:
        if key.endswith('bias'):
            return torch.randn(shape, generator=g) * 0.1
        return torch.rand(shape, generator=g) * 0.4 + 0.8               # norm scale
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    # gain 1 keeps the 23-block encoder O(1) without BN calibration; the hypernetwork heads get gain 4 so that
    # the dynamic weights (not the BN biases) decide the argmax.  Checked: fp32-vs-fp64 error of the whole
    # HyperSeg-M forward with these weights is 7e-7 (a He-gain variant was ill-conditioned: 1.5e-3).
    gain = 4.0 if 'signal2weights' in key else 1.0
    return torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)   # conv / linear weight


def fill_by_name(module, seed=0):
    sd = module.state_dict()
    # cached coordinate buffers (decoder.coord{h}_{w}) keep their constructed linspace values: they are data the
    # constructor defines, not weights (the HIP path regenerates them analytically, Appendix D-11)
    new = {k: (tensor_for(k, tuple(v.shape), seed).to(v.dtype)
               if v.dtype.is_floating_point and '.coord' not in k else v) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module
