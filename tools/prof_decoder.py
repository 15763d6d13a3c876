"""Run the decoder N times (eager) for rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from oracle import hyperseg_oracle as O
from test_hip_parity import build_decoder
name = sys.argv[1] if len(sys.argv) > 1 else 'M'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device('cuda:0')
d = build_decoder(name, O).to(dev)
x, s = O.synth_decoder_inputs(name, batch=1, seed=0)
x = [t.to(dev) for t in x]; s = s.to(dev)
torch.set_grad_enabled(False)
for _ in range(n):
    d(x, s)
torch.cuda.synchronize()
