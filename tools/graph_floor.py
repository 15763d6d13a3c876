"""Floor of a dependent launch inside a replayed HIP graph on this box: N tiny elementwise kernels chained on one tensor,
captured once, replayed; us per node for a 1-element, a 64 K-element and a 1 M-element tensor.  The encoder's ~200
launches per frame average 4.9 us each (profiles/round2_bench_kernel_stats.csv) -- this says how much of that is floor.
    python tools/graph_floor.py [nodes]"""
import sys
import time
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
for numel in (1, 1 << 16, 1 << 20):
    x = torch.zeros(numel, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            x.add_(1.0)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    reps = 50
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f'{numel:8d} elements: {el / reps / n * 1e6:6.2f} us per dependent node ({n} nodes per graph, {reps} replays)')
