"""Dev diagnostic (round 2): which component makes the CamVid-S level-3 train-mode gradients deviate at 576x576 bs2?
Runs the level-3 inverted residual alone (24 -> 48 -> 16 at 144x144, grid 18x18, batch 2) on the GPU against the CPU
oracle, with MIOpen batch norm on/off, and the raw patch-conv gradients at the same shapes."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import hyperseg_oracle as O          # dev diagnostic only (not product code)
from hyperseg_amd.models import hyperseg_v1_0 as M
from hyperseg_amd.models.layers.meta_patch import MetaPatchConv2d

dev = torch.device('cuda:0')
rel = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max())
g = torch.Generator().manual_seed(0)
B, cin, hid, cout, H, fh = 2, 24, 48, 16, 144, 18


def block_case(cudnn):
    torch.backends.cudnn.enabled = cudnn
    m = M.HyperPatchInvertedResidual(cin, cout, 3, expand_ratio=2)
    hp = m.hyper_params
    gg = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, H, H, generator=gg)
    wt = torch.randn(B, hp, fh, fh, generator=gg)
    wt[:, :cin * hid] *= (2.0 / cin) ** 0.5
    wt[:, cin * hid:cin * hid + 9 * hid] *= (2.0 / 9) ** 0.5
    wt[:, cin * hid + 9 * hid:] *= (1.0 / hid) ** 0.5
    r = torch.randn(B, cout, H, H, generator=gg)
    bns = [{k: getattr(bn, k).detach().clone() for k in ('weight', 'bias', 'running_mean', 'running_var')}
           for bn in (m.bn1, m.bn2, m.bn3)]
    for d in bns:
        d['weight'].requires_grad_(True); d['bias'].requires_grad_(True)
    xo, wo = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yo, _ = O.patch_inverted_residual_v1(xo, wo, hid, cout, *bns, training=True)
    (yo * r).sum().backward()
    m = m.to(dev).train()
    xg, wg = x.to(dev).requires_grad_(True), wt.to(dev).requires_grad_(True)
    yg = m.conv(xg, wg)
    (yg * r.to(dev)).sum().backward()
    print(f'cudnn/MIOpen BN = {cudnn}: y {rel(yg.detach(), yo.detach()):.1e} dx {rel(xg.grad, xo.grad):.1e} dw {rel(wg.grad, wo.grad):.1e} '
          + ' '.join(f'bn{i+1}.w {rel(getattr(m, f"bn{i+1}").weight.grad, bns[i]["weight"].grad):.1e} bn{i+1}.b {rel(getattr(m, f"bn{i+1}").bias.grad, bns[i]["bias"].grad):.1e}'
                     for i in range(3)))


def conv_case(ci, co, k, groups, mode, patch, tag):
    gg = torch.Generator().manual_seed(2)
    h = fh * patch
    m = MetaPatchConv2d(ci, co, k, padding=k // 2, groups=groups, padding_mode=mode)
    x = torch.randn(B, ci, h, h, generator=gg)
    wt = torch.randn(B, m.hyper_params, fh, fh, generator=gg)
    r = torch.randn(B, co, h, h, generator=gg)
    xo, wo = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    yo = O.meta_patch_conv2d(xo, wo, co, k, k // 2, mode, groups)
    (yo * r).sum().backward()
    xg, wg = x.to(dev).requires_grad_(True), wt.to(dev).requires_grad_(True)
    yg = m(xg, wg)
    (yg * r.to(dev)).sum().backward()
    print(f'{tag}: y {rel(yg.detach(), yo.detach()):.1e} dx {rel(xg.grad, xo.grad):.1e} dw {rel(wg.grad, wo.grad):.1e}')


conv_case(48, 16, 1, 1, 'zeros', 8, 'pw3 48->16 @144 p8')
conv_case(24, 48, 1, 1, 'zeros', 10, 'pw1 24->48 @180 tile10')
conv_case(48, 48, 3, 48, 'zeros', 10, 'dw 48 @180 tile10')
block_case(True)
block_case(False)
