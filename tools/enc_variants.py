"""Graph-replay frame time of HyperSeg-M under encoder preparation variants (GPU box).  Dev tool."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
from hyperseg_amd.utils.inference import prepare_for_inference
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
x0 = torch.rand(1, 3, 512, 1024, device=dev)
ref = None
for variant in sys.argv[1:] or ['plain', 'fold', 'cl', 'foldcl', 'foldbench']:
    torch.backends.cudnn.benchmark = 'bench' in variant
    m = fill_by_name(configs.build('hyperseg-m').eval(), seed=0)
    n = prepare_for_inference(m, fold_bn='fold' in variant, channels_last='cl' in variant, fused_depthwise='dw' in variant)
    m = m.to(dev)
    x = x0.contiguous(memory_format=torch.channels_last) if 'cl' in variant else x0
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y = m(x)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = m(x)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        g.replay()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 10
    if ref is None:
        ref = y.clone()
    print(f'{variant:10s} folded {n:3d}  {ms:.3f} ms/frame = {1e3/ms:.1f} FPS   max|dy|/max|y| vs plain {float((y-ref).abs().max()/ref.abs().max()):.2e}', flush=True)
