"""Whole HyperSeg-M forward, eager, for rocprofv3 --kernel-trace --stats.  python tools/prof_model.py [n] [variant]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variant = sys.argv[2] if len(sys.argv) > 2 else 'plain'
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = fill_by_name(configs.build('hyperseg-m').eval(), seed=0).to(dev)
if 'bench' in variant:
    torch.backends.cudnn.benchmark = True
if 'cl' in variant:
    m.backbone = m.backbone.to(memory_format=torch.channels_last)
    m.weight_mapper = m.weight_mapper.to(memory_format=torch.channels_last)
x = torch.rand(1, 3, 512, 1024, device=dev)
if 'cl' in variant:
    x = x.contiguous(memory_format=torch.channels_last)
for _ in range(n):
    m(x)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(n):
    m(x)
torch.cuda.synchronize()
print(variant, 'eager ms/frame', (time.perf_counter() - t0) * 1e3 / n)
