"""Developer probe (GPU box): per-launch timings of the decoder at a BASELINE config.
    python tools/gpu_probe.py [M|Sc] [iters]
Not part of the product or the tests."""
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hyperseg_amd.functional as HF
from oracle import hyperseg_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'M'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    from test_hip_parity import build_decoder
    dev = torch.device('cuda:0')
    d = build_decoder(name, O).to(dev)
    x, s = O.synth_decoder_inputs(name, batch=1, seed=0)
    x = [t.to(dev) for t in x]
    s = s.to(dev)
    torch.set_grad_enabled(False)
    for _ in range(3):
        y = d(x, s)
    torch.cuda.synchronize()

    # per-launch timing: wrap the functional entry points
    records = {}
    names = ['signal2weights', 'signal2weights_multi', 'bank_pack', 'patch_conv', 'patch_ir', 'upsample_bilinear']
    orig = {n: getattr(HF, n) for n in names}
    counter = [0]

    def wrap(n):
        def f(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig[n](*a, **k)
            e1.record()
            records.setdefault((counter[0], n), []).append((e0, e1))
            counter[0] += 1
            return r
        return f
    for n in names:
        setattr(HF, n, wrap(n))
    for _ in range(iters):
        counter[0] = 0
        d(x, s)
    torch.cuda.synchronize()
    for n in names:
        setattr(HF, n, orig[n])
    tot = 0.0
    for (i, n), evs in sorted(records.items()):
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        med = ts[len(ts) // 2]
        tot += med
        print(f'launch {i:2d} {n:18s} median {med:8.1f} us   min {ts[0]:8.1f} us')
    print(f'sum of medians {tot:.1f} us')

    # whole decoder: eager and HIP graph
    def timeit(fn, n):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n, (time.perf_counter() - t0) * 1e6 / n
    dev_us, wall_us = timeit(lambda: d(x, s), iters)
    print(f'eager decoder: {dev_us:.1f} us/frame (events), {wall_us:.1f} us wall')
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for _ in range(3):
            d(x, s)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(g):
        yg = d(x, s)
    dev_us, wall_us = timeit(g.replay, iters * 4)
    print(f'graph  decoder: {dev_us:.1f} us/frame (events), {wall_us:.1f} us wall')
    ref = O.run_config(name, batch=1, seed=0)
    print('graph output rel err vs oracle:', float((yg.cpu() - ref).abs().max() / ref.abs().max()))


if __name__ == '__main__':
    main()
