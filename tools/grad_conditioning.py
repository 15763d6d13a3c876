"""How well-conditioned is the config-5 training gradient?  (CPU, fp64 oracle; evidence for the tolerance of
tests/test_hip_training.py::test_config5_full_workload_fp32.)  Perturbs inputs and parameters of the CamVid-S decoder at
576x576 bs2 by one fp32 ulp (6e-8 relative) and reports how far the fp64 oracle's own gradients move.
    python tools/grad_conditioning.py            (~4 minutes)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hyperseg_oracle as O          # dev tool for the TEST tolerance: the oracle is the object of study here


def run(eps=0.0, seed=0):
    dtype = torch.float64
    plan = O.config_plan('Sc')
    params = O.synth_decoder_params(plan, seed=0)
    x, s = O.synth_decoder_inputs('Sc', batch=2, seed=3, size=(576, 576))
    r = torch.randn(2, 12, 576, 576, generator=torch.Generator().manual_seed(4))
    g = torch.Generator().manual_seed(100 + seed)
    pert = (lambda t: t.to(dtype) * (1 + eps * torch.randn(t.shape, generator=g, dtype=dtype))) if eps else (lambda t: t.to(dtype))
    po = {k: (pert(v).clone().requires_grad_(True) if v.dtype.is_floating_point and 'running' not in k else
              (v.to(dtype) if v.dtype.is_floating_point else v.clone())) for k, v in params.items()}
    xo = [pert(t).requires_grad_(True) for t in x]
    so = pert(s).requires_grad_(True)
    y, _ = O.decoder_v1_0(plan, po, xo, so, training=True)
    (y * r.to(dtype)).sum().backward()
    out = {'logits': y.detach(), 'd signal': so.grad}
    for i in range(1, 6):
        out[f'd pyramid[{i}]'] = xo[i].grad
    for k, v in po.items():
        if v.requires_grad and v.grad is not None:
            out['d ' + k] = v.grad
    return out


if __name__ == '__main__':
    base = run()
    for seed in (0, 1):
        p = run(eps=6e-8, seed=seed)
        print(f'--- perturbation seed {seed}: tensor, max-norm error, relative L2 error')
        for k in base:
            d = (p[k] - base[k]).abs()
            print(f'{k:42s} {float(d.max() / base[k].abs().max()):.1e}  {float(d.norm() / base[k].norm()):.1e}')
