"""Config-5 training step replayed as one HIP graph, for rocprofv3 --kernel-trace: a marker scan kernel separates the warm-up from the N
measured replays (tools/frame_sequence.py reads the trace: in-order per-launch durations, gaps and grids of one step).
    python tools/prof_train_graph.py [n_replays] [fp32|bf16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.training import Adam, BootstrappedCrossEntropyLoss, GraphedTrainStep
from hyperseg_amd.utils.synthetic import fill_by_name

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
half = len(sys.argv) > 2 and sys.argv[2] == 'bf16'
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
opt = Adam(dec.parameters(), lr=torch.tensor(1e-3, device=dev), betas=(0.5, 0.999))


def fwd(p, sig):
    with torch.autocast('cuda', dtype=torch.bfloat16, enabled=half):
        return dec(p, sig)


gs = GraphedTrainStep(fwd, crit, opt, (pyr, s), target)
for _ in range(3):
    gs.step()
torch.cuda.synchronize()
marker = torch.cumsum(torch.ones(4096, device=dev), 0)       # marker kernel (a scan: not used by the step)
torch.cuda.synchronize()
for _ in range(n):
    gs.step()
torch.cuda.synchronize()
