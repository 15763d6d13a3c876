"""Runs one BASELINE configuration's decoder N times on resident synthetic inputs (for rocprofv3 --kernel-trace --stats).
    python tools/decoder_loop.py M|S|Sc|L [iters] [batch]"""
import sys
import time
import torch
from _workload import decoder_workload

name = sys.argv[1] if len(sys.argv) > 1 else 'M'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
batch = int(sys.argv[3]) if len(sys.argv) > 3 else None
dec, pyr, head = decoder_workload(name, batch=batch)
for _ in range(3):
    dec(pyr, head)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    y = dec(pyr, head)
torch.cuda.synchronize()
print(f'{name} decoder eager: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/batch, out {tuple(y.shape)}')
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = dec(pyr, head)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    g.replay()
torch.cuda.synchronize()
print(f'{name} decoder graph replay: {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/batch')
