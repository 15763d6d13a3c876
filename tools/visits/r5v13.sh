#!/bin/bash
# round 5, visit 13: where PatchConvBN differs from BNActTrain + PatchConv in fp32 (per-key max differences)
tag=${1:-r5v13}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_bn_diag_$tag.txt
import copy, torch, torch.nn as nn
from hyperseg_amd import autograd as HA
dev = torch.device('cuda:0')
for (c, cout, grid, p) in [(44, 12, (3, 3), 16), (48, 16, (3, 2), 8)]:
    g = torch.Generator().manual_seed(1)
    b = 2; fh, fw = grid; h, w = fh * p, fw * p
    x0 = (torch.randn(b, c, h, w, generator=g) * 1.3 + 0.5).to(dev)
    bank0 = (torch.randn(b * fh * fw, cout * c, generator=g) / c ** 0.5).to(dev)
    r = torch.randn(b, cout, h, w, generator=g).to(dev)
    bn0 = nn.BatchNorm2d(c).to(dev).train()
    def run(fused):
        HA.USE_CONV_BN_FUSED = fused
        bn = copy.deepcopy(bn0); x = x0.clone().requires_grad_(True); bank = bank0.clone().requires_grad_(True)
        y = HA.patch_conv_bn(bn, nn.ReLU6(), x, bank, grid, cout)
        (y * r).sum().backward()
        return dict(y=y.detach(), dx=x.grad, dbank=bank.grad, dg=bn.weight.grad, db=bn.bias.grad, rm=bn.running_mean.clone(), rv=bn.running_var.clone())
    one, two = run(True), run(False)
    # the normalised map of the two-step route, and the conv of it by the fused route's kernel choice
    bn = copy.deepcopy(bn0)
    z = HA.bn_act(bn, nn.ReLU6(), x0.clone())
    y2 = HA.patch_conv_apply(z, bank0, grid, cout, 1, 0, 'zeros', 1)
    print((c, cout, grid, p), {k: float((one[k] - two[k]).abs().max()) for k in one}, 'y two-step again', float((y2 - two['y']).abs().max()),
          'nonzero frac', float(((one['y'] - two['y']) != 0).float().mean()))
    d = (one['y'] - two['y']).abs()
    idx = d.flatten().argmax().item()
    print('   at', idx, 'one', one['y'].flatten()[idx].item(), 'two', two['y'].flatten()[idx].item())
PY
