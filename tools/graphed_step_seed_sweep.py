"""Seed sweep behind tests/test_hip_training.py::test_graphed_train_step_equals_eager (a 1-in-25 failure of that
test had been pinned by seeding, not explained).

    python tools/graphed_step_seed_sweep.py [n_seeds=200] [steps=5]  > profiles/round5_graphed_step_seed_sweep.txt

Per target seed, three CamVid-S decoders start from the SAME state and take ``steps`` training steps on the same batch:
  P  eager, plain ``torch.optim.Adam``                       (what round 4's test compared the replays with)
  C  eager, ``Adam(capturable=True, lr=tensor)``             (the optimizer the captured step has to use)
  G  hyperseg_amd.training.GraphedTrainStep replays of C's step (captured ONCE; per seed the parameters, BatchNorm buffers and
     optimizer state are reset in place, so every seed replays the same graph)
and the table shows, per pair, the largest relative loss gap over the steps and the largest / mean absolute parameter gap after them.
G vs C isolates the capture (same arithmetic: any gap is nondeterminism or a capture bug); C vs P isolates the optimizer's
bias-correction arithmetic (device-side ``step`` tensor vs host float), whose last-bit differences Adam turns into a sign flip of
an update -- i.e. a gap of up to 2 lr per step -- wherever a gradient component sits at rounding-noise level."""
import copy
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    from oracle import hyperseg_oracle as O                       # test infrastructure: synthetic CamVid-S weights and inputs
    from test_hip_parity import build_decoder
    from hyperseg_amd.training import BootstrappedCrossEntropyLoss, GraphedTrainStep
    dev = torch.device('cuda:0')
    lr = 1e-3
    base = build_decoder('Sc', O).to(dev).train()
    state0 = copy.deepcopy(base.state_dict())
    x, s = O.synth_decoder_inputs('Sc', batch=2, seed=3, size=(96, 96))
    x, s = [t.to(dev) for t in x], s.to(dev)
    crit = BootstrappedCrossEntropyLoss(k=512, thresh=0.3, ignore_index=255)
    target = torch.randint(0, 12, (2, 96, 96), generator=torch.Generator().manual_seed(0)).to(dev)

    dP, dC, dG = copy.deepcopy(base), copy.deepcopy(base), copy.deepcopy(base)
    oC = torch.optim.Adam(dC.parameters(), lr=torch.tensor(lr, device=dev), betas=(0.5, 0.999), capturable=True)
    oG = torch.optim.Adam(dG.parameters(), lr=torch.tensor(lr, device=dev), betas=(0.5, 0.999), capturable=True)
    gs = GraphedTrainStep(dG, crit, oG, (x, s), target, warmup=2)

    def reset(model, opt):
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(state0[k])
            if opt is not None:
                for st in opt.state.values():
                    for v in st.values():
                        if isinstance(v, torch.Tensor):
                            v.zero_()

    def eager(model, opt, tgt):
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            loss = crit(model(x, s), tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
            del loss
        return losses

    def gaps(a, b):
        sa, sb = a.state_dict(), b.state_dict()
        pmax = pmean = rstat = 0.0
        for k in sa:
            if not sa[k].dtype.is_floating_point:
                assert torch.equal(sa[k], sb[k]), k
                continue
            d = (sa[k] - sb[k]).abs()
            if 'running_' in k:
                rstat = max(rstat, float(d.max() / sb[k].abs().max().clamp(min=1e-30)))
            else:
                pmax, pmean = max(pmax, float(d.max())), max(pmean, float(d.mean()))
        return pmax, pmean, rstat

    print(f'# graphed_step_seed_sweep: {n_seeds} target seeds x {steps} steps, CamVid-S decoder 96x96 bs 2, Adam lr {lr} betas (0.5, 0.999); '
          f'{torch.cuda.get_device_name(0)}')
    print('# G = GraphedTrainStep replays, C = eager capturable Adam, P = eager plain Adam; loss gaps relative, parameter gaps in units of lr')
    print('seed   loss G-C    pmax G-C  pmean G-C  stat G-C |  loss C-P    pmax C-P  pmean C-P  stat C-P')
    rows = []
    for seed in range(n_seeds):
        tgt = torch.randint(0, 12, (2, 96, 96), generator=torch.Generator().manual_seed(1000 + seed)).to(dev)
        reset(dP, None)
        oP = torch.optim.Adam(dP.parameters(), lr=lr, betas=(0.5, 0.999))
        lP = eager(dP, oP, tgt)
        reset(dC, oC)
        lC = eager(dC, oC, tgt)
        reset(dG, oG)
        lG = [float(gs.step(target=tgt)[0]) for _ in range(steps)]
        torch.cuda.synchronize()
        gl = max(abs(a - b) / abs(b) for a, b in zip(lG, lC))
        cl = max(abs(a - b) / abs(b) for a, b in zip(lC, lP))
        g, c = gaps(dG, dC), gaps(dC, dP)
        rows.append((gl, g[0] / lr, g[1] / lr, g[2], cl, c[0] / lr, c[1] / lr, c[2]))
        print(f'{seed:4d}  {gl:9.2e}  {g[0] / lr:9.2e}  {g[1] / lr:9.2e} {g[2]:9.2e} | {cl:9.2e}  {c[0] / lr:9.2e}  {c[1] / lr:9.2e} {c[2]:9.2e}')
    t = torch.tensor(rows, dtype=torch.float64)
    names = ['loss G-C', 'pmax G-C [lr]', 'pmean G-C [lr]', 'stat G-C', 'loss C-P', 'pmax C-P [lr]', 'pmean C-P [lr]', 'stat C-P']
    print('\n# distribution over seeds:           median        p90        p99        max')
    for i, nme in enumerate(names):
        col = t[:, i]
        q = torch.quantile(col, torch.tensor([0.5, 0.9, 0.99], dtype=torch.float64))
        print(f'# {nme:<28s} {float(q[0]):10.3e} {float(q[1]):10.3e} {float(q[2]):10.3e} {float(col.max()):10.3e}')


if __name__ == '__main__':
    main()
