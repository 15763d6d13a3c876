"""Which host-side ops issue the stock launches of a config-5 training step: torch.profiler over two eager steps, device kernels /
memcpys grouped by the CPU op that launched them (name of the innermost aten / autograd op).
    python tools/train_step_trace.py [fp32|bf16]"""
import collections
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from hyperseg_amd import configs
from hyperseg_amd.training import BootstrappedCrossEntropyLoss
from hyperseg_amd.utils.synthetic import fill_by_name

mode = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
dev = torch.device('cuda:0')
model = fill_by_name(configs.build('hyperseg-s-camvid'), seed=0).to(dev)
x = torch.rand(2, 3, 576, 576, device=dev)
with torch.no_grad():
    model.eval()
    feats = model.backbone(x)
    s = model.weight_mapper(feats[-1]).contiguous()
    pyr = [t.contiguous() for t in [x] + feats[:-1]]
dec = model.decoder.train()
target = torch.randint(0, 12, (2, 576, 576), device=dev)
crit = BootstrappedCrossEntropyLoss(k=4096, thresh=0.3, ignore_index=255)
opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999), fused=True)


def step():
    opt.zero_grad()
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=(mode == 'bf16')):
        pred = dec(pyr, s)
    loss = crit(pred.float(), target)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
ev = prof.events()
cpu = [e for e in ev if e.device_type == torch.autograd.DeviceType.CPU]
by_op = collections.Counter()
names = collections.defaultdict(collections.Counter)
for e in cpu:
    if not e.kernels:
        continue
    # innermost op = the event itself if none of its children own kernels
    if any(c.kernels for c in e.cpu_children):
        continue
    for k in e.kernels:
        kn = k.name[:60]
        if 'hs::' in kn[:12]:
            continue
        by_op[e.name] += 1
        names[e.name][kn] += 1
print(f'{mode}: stock device launches of 2 steps by launching op')
for op, n in by_op.most_common(30):
    print(f'{n / 2:6.1f}/step  {op[:70]}   <- ' + ', '.join(f'{c}x {k[:40]}' for k, c in names[op].most_common(2)))
# parents of aten::copy_ / clone
par = collections.Counter()
for e in cpu:
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::fill_', 'aten::zero_', 'aten::zeros') and e.kernels or \
            (e.name in ('aten::copy_', 'aten::fill_') and any(True for _ in e.kernels)):
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name[:50])
            p = p.cpu_parent
        par[(e.name, ' < '.join(chain))] += 1
print('callers of the copies / fills:')
for (n, chain), c in par.most_common(40):
    print(f'{c / 2:6.1f}/step  {n:18s} {chain}')
