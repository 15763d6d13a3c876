"""Whole-model HIP-graph replays for rocprofv3 --kernel-trace; a marker fill kernel (size 77777) separates the warm-up
from the N measured replays.  python tools/prof_graph.py [n_replays] [fold|plain]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
from hyperseg_amd.utils.inference import prepare_for_inference
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
variant = sys.argv[2] if len(sys.argv) > 2 else 'plain'
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
m = fill_by_name(configs.build('hyperseg-m').eval(), seed=0)
prepare_for_inference(m, fold_bn='fold' in variant, fused_depthwise='dw' in variant, split_gemm='dw' in variant)
m = m.to(dev)
x = torch.rand(1, 3, 512, 1024, device=dev)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        m(x)
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = m(x)
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
marker = torch.cumsum(torch.ones(4096, device=dev), 0)       # marker kernel (a scan: not used by the model)
torch.cuda.synchronize()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
