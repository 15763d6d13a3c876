"""Time of one config-5 training step of the decoder (CamVid-S, 576x576 crops, bs 2: forward + loss + backward + Adam)
through the HIP path, fp32 and bf16 autocast, plus the per-kernel picture under rocprofv3 if wrapped.
    python tools/train_step_time.py [iters] [fp32|bf16]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.training import BootstrappedCrossEntropyLoss
from hyperseg_amd.utils.synthetic import fill_by_name

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
model = fill_by_name(configs.build('hyperseg-s-camvid'), seed=0).to(dev)
x = torch.rand(2, 3, 576, 576, device=dev)
with torch.no_grad():
    model.eval()
    feats = model.backbone(x)
    s = model.weight_mapper(feats[-1]).contiguous()
    pyr = [t.contiguous() for t in [x] + feats[:-1]]
if os.environ.get('HS_TRAIN_SIGNAL_GRAD') == '1':       # as inside a whole-model step: the signal is the context head's output and needs its gradient
    s.requires_grad_(True)
dec = model.decoder.train()
target = torch.randint(0, 12, (2, 576, 576), device=dev)
crit = BootstrappedCrossEntropyLoss(k=4096, thresh=0.3, ignore_index=255)
FUSED = os.environ.get('HS_ADAM_FUSED', '1') == '1'      # torch's single-launch Adam (same arithmetic as the foreach form the reference's optim.Adam resolves to)
OURS = os.environ.get('HS_ADAM', 'ours') == 'ours'        # hyperseg_amd.training.Adam (one launch of 1024-element workgroups); 'torch': torch.optim.Adam
if os.environ.get('HS_SHARED_BANK_GRAD') == '0':
    import hyperseg_amd.autograd as _HA
    _HA.USE_SHARED_BANK_GRAD = False
from hyperseg_amd.training import Adam as OwnAdam
opt = OwnAdam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999)) if OURS else torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999), fused=FUSED)
modes = sys.argv[2:] if len(sys.argv) > 2 else ('fp32', 'bf16', 'graph', 'graph_bf16')
for mode in modes:
    if mode in ('graph', 'graph_bf16'):                     # the step captured once and replayed (hyperseg_amd.training.GraphedTrainStep)
        from hyperseg_amd.training import GraphedTrainStep
        opt_g = OwnAdam(dec.parameters(), lr=torch.tensor(1e-3, device=dev), betas=(0.5, 0.999)) if OURS else \
            torch.optim.Adam(dec.parameters(), lr=torch.tensor(1e-3, device=dev), betas=(0.5, 0.999), capturable=True, fused=FUSED)

        def fwd(p, sig, half=(mode == 'graph_bf16')):
            with torch.autocast('cuda', dtype=torch.bfloat16, enabled=half):
                return dec(p, sig)
        gs = GraphedTrainStep(fwd, crit, opt_g, (pyr, s), target)
        for _ in range(2):
            gs.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, _ = gs.step()
        torch.cuda.synchronize()
        print(f'config-5 decoder training step ({"bf16 autocast" if mode == "graph_bf16" else "fp32"}, one HIP graph per step): {(time.perf_counter() - t0) / iters * 1e3:.2f} ms/step, loss {float(loss):.4f}')
        continue
    def step():
        opt.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16, enabled=(mode == 'bf16')):
            pred = dec(pyr, s)
        loss = crit(pred, target)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = step()
    torch.cuda.synchronize()
    print(f'config-5 decoder training step ({mode}): {(time.perf_counter() - t0) / iters * 1e3:.2f} ms/step, loss {float(loss):.4f}')
    del loss                                                # a live autograd graph must not outlive the eager modes (GraphedTrainStep docstring)
