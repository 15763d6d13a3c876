import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
torch.set_grad_enabled(False)
m = fill_by_name(configs.build('hyperseg-m').eval(), seed=11)
x = torch.rand(1, 3, 128, 256, generator=torch.Generator().manual_seed(12))
fc = m.backbone(x); sc = m.weight_mapper(fc[-1])
mg = m.to('cuda:0')
fg = mg.backbone(x.cuda()); sg = mg.weight_mapper(fg[-1])
for i,(a,b) in enumerate(zip(fc,fg)):
    print('feat',i,tuple(a.shape), float(a.abs().max()), 'rel err', float((a-b.cpu()).abs().max()/a.abs().max()))
print('signal', float(sc.abs().max()), float((sc-sg.cpu()).abs().max()/sc.abs().max()))
# block by block
mc = fill_by_name(configs.build('hyperseg-m').eval(), seed=11).backbone
bg = mg.backbone
import torch.nn.functional as F
a = F.silu(mc._bn0(mc._conv_stem(x))); b = F.silu(bg._bn0(bg._conv_stem(x.cuda())))
print('stem', float((a-b.cpu()).abs().max()/a.abs().max()))
for i,(blc,blg) in enumerate(zip(mc._blocks,bg._blocks)):
    a2 = blc(a); b2 = blg(a.cuda())      # same input to both
    print('block',i,'absmax',float(a2.abs().max()),'rel err (same input)', float((a2-b2.cpu()).abs().max()/a2.abs().max()))
    a = a2
# decoder given identical CPU features
from oracle import hyperseg_oracle as O
plan = O.config_plan('M')
params = {k: v.cpu() for k, v in mg.decoder.state_dict().items()}
pyr = [x] + fc[:-1]
ref = O.decoder_v1_0(plan, params, pyr, sc)
y = mg.decoder([t.cuda().contiguous() for t in pyr], sc.cuda().contiguous()).cpu()
print('decoder on identical inputs: absmax', float(ref.abs().max()), 'rel err', float((y-ref).abs().max()/ref.abs().max()))
import numpy as np
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests/golden/model_M.npz'))
yfull = mg(x.cuda()).cpu()
print('whole model vs golden sample: abs err', float((yfull[:, :, 1::5, 2::7] - torch.from_numpy(g['y'])).abs().max()), 'absmax', float(g['y_absmax']))
print('whole model vs (cpu feats + oracle):', float((yfull - ref).abs().max()))
print('oracle(ref) vs golden sample:', float((ref[:, :, 1::5, 2::7] - torch.from_numpy(g['y'])).abs().max()))
