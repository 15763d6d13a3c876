"""Dev probe: blocked vs direct signal2weights on one layer shape -- where do they differ?
    python tools/s2w_ab.py cs groups rows fh fw [batch] [sidx]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import functional as HF
cs, groups, rows, fh, fw = map(int, sys.argv[1:6])
batch = int(sys.argv[6]) if len(sys.argv) > 6 else 1
sidx = int(sys.argv[7]) if len(sys.argv) > 7 else 0
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
wc = -(-rows // groups) * groups
s = torch.relu(torch.randn(batch, sidx + cs + 5, fh, fw, generator=g)).to(dev)
wsw = torch.randn(wc, cs // groups, generator=g)
layer = dict(wsw_t=wsw.t().contiguous().to(dev), signal_index=sidx, signal_channels=cs, groups=groups, rows=rows)
HF.S2W_BLOCKED = False
a = HF.signal2weights_multi(s, [layer])[0].bank.clone()
HF.S2W_BLOCKED = True
b = HF.signal2weights_multi(s, [layer])[0].bank.clone()
torch.cuda.synchronize()
d = (a[:, :rows] - b[:, :rows]).abs()
print('max diff', float(d.max()), 'of', float(a.abs().max()))
bad = (d > 1e-6).nonzero()
print('bad entries', len(bad))
if len(bad):
    print('patches', sorted(set(bad[:, 0].tolist()))[:40])
    ns = sorted(set(bad[:, 1].tolist()))
    print('rows', ns[:20], '...', ns[-5:], 'count', len(ns))
