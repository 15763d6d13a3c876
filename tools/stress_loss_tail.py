"""Stress of the bootstrapped loss' in-launch tail (round 6: the last-arriving workgroup of bm_sums_kernel finishes every image; partial sums leave as
write-through stores, each workgroup takes a ticket after its stores are acknowledged): N calls of the fused loss against the two-Function route on
fresh random inputs, eager and replayed from a HIP graph -- every loss and gradient must be bit-equal.
    python tools/stress_loss_tail.py [n_calls]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hyperseg_amd.training as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device('cuda:0')
g = torch.Generator(device='cpu').manual_seed(0)
bad = 0
for it in range(n):
    b = 1 + it % 3
    h, w = 96 + 8 * (it % 5), 160 + 16 * (it % 7)
    x = (torch.randn(b, 12, h, w, generator=g) * (0.5 + it % 4)).to(dev)
    t = torch.randint(0, 12, (b, h, w), generator=g).to(dev)
    t[:, : it % 6] = 255
    k, thresh = 512 + 37 * (it % 11), (0.3, 1.5, 4.0)[it % 3]
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    T.USE_FUSED_LOSS = True
    la = T.bootstrapped_cross_entropy(xa, t, k=k, thresh=thresh, ignore_index=255)
    T.USE_FUSED_LOSS = False
    lb = T.bootstrapped_cross_entropy(xb, t, k=k, thresh=thresh, ignore_index=255)
    T.USE_FUSED_LOSS = True
    la.backward(); lb.backward()
    ref = T.bootstrap_mean_reference if hasattr(T, 'bootstrap_mean_reference') else None
    if not (torch.equal(la, lb) and torch.equal(xa.grad, xb.grad)):
        bad += 1
        print('MISMATCH at call', it, float(la), float(lb))
# replayed: one captured loss forward + backward, new logits copied in before each replay
x = torch.randn(2, 12, 288, 288, device=dev).requires_grad_(True)
t = torch.randint(0, 12, (2, 288, 288), device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        x.grad = None
        T.bootstrapped_cross_entropy(x, t, k=4096, thresh=0.3, ignore_index=255).backward()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
x.grad = None
with torch.cuda.graph(graph):
    loss = T.bootstrapped_cross_entropy(x, t, k=4096, thresh=0.3, ignore_index=255)
    loss.backward()
for it in range(n):
    with torch.no_grad():
        x.copy_(torch.randn(x.shape, generator=g).to(dev) * (0.5 + it % 4))
    graph.replay()
    lg, gg = loss.clone(), x.grad.clone()
    xe = x.detach().clone().requires_grad_(True)
    T.USE_FUSED_LOSS = False
    le = T.bootstrapped_cross_entropy(xe, t, k=4096, thresh=0.3, ignore_index=255)
    T.USE_FUSED_LOSS = True
    le.backward()
    if not (torch.equal(lg, le) and torch.equal(gg, xe.grad)):
        bad += 1
        print('REPLAY MISMATCH at', it, float(lg), float(le))
print(f'stress_loss_tail: {2 * n} calls, {bad} mismatches')
sys.exit(1 if bad else 0)
