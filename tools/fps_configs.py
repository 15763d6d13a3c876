"""Frames/s of the other BASELINE configurations (HyperSeg-S 1536x768 bs1, HyperSeg-S CamVid 768x576,
HyperSeg-L 512x512 bs32) with the same protocol as bench.py (resident synthetic batch, whole model, HIP-graph replay) plus
the decoder alone.  One JSON line per configuration.   python tools/fps_configs.py [names...]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
from hyperseg_amd.utils.inference import prepare_for_inference

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
names = sys.argv[1:] or ['hyperseg-s', 'hyperseg-s-camvid', 'hyperseg-l', 'hyperseg-m']


def replay_ms(fn, iters):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / iters


for name in names:
    spec = configs.MODELS[name]
    h, w = spec['size']
    b = spec['batch']
    m = fill_by_name(configs.build(name).eval(), seed=0)
    prepare_for_inference(m, fold_bn=False, fused_depthwise=True)
    m = m.to(dev)
    x = torch.rand(b, 3, h, w, device=dev)
    feats = m.backbone(x)
    head = m.weight_mapper(feats[-1])
    head = head.contiguous() if isinstance(head, torch.Tensor) else head
    pyr = [t.contiguous() for t in [x] + feats[:-1]]
    iters = 100 if b == 1 else 20
    ms_model = replay_ms(lambda: m(x), iters)
    ms_dec = replay_ms(lambda: m.decoder(pyr, head), iters)
    ms_seg = replay_ms(lambda: m.segment(x), iters)
    print(json.dumps({'config': name, 'batch': b, 'size': [h, w], 'ms_per_batch_model': round(ms_model, 3),
                      'frames_per_s_model': round(b / ms_model * 1e3, 1), 'ms_per_batch_decoder': round(ms_dec, 3),
                      'ms_per_batch_segment': round(ms_seg, 3), 'frames_per_s_segment': round(b / ms_seg * 1e3, 1)}), flush=True)
    del m, x, feats, head, pyr
    torch.cuda.empty_cache()
