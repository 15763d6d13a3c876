"""Per-phase cycle counts of k1m_pixel_stream (hs_patch_conv_bwd.hip) from the 'stamps_k1m' dev build, on config 5's level-4 pw1 forward
(22 -> 44 channels on 648 halo tiles of 18 x 18) or level 3's (24 -> 48 on 10 x 10):
    HS_HIP_LIB=hyperseg_amd/lib/libhyperseg_hip_stamps_k1m.so python tools/k1m_phase_times.py [4|3]"""
import ctypes as C
import collections
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hyperseg_amd._hip as hip
from hyperseg_amd import functional as HF

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cin, cout, tile = (22, 44, 18) if level == 4 else (24, 48, 10)
b, fh, fw = 2, 18, 18
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = torch.randn(b, cin, fh * tile, fw * tile, generator=g).to(dev)
bank = torch.randn(b * fh * fw, cin * cout, generator=g).to(dev)
for _ in range(5):
    y = HF.patch_conv(x, (fh, fw), bank, cout, 1, 0, 'zeros', 1)
torch.cuda.synchronize()
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    y = HF.patch_conv(x, (fh, fw), bank, cout, 1, 0, 'zeros', 1)
t1.record(); torch.cuda.synchronize()
print(f'level {level}: {t0.elapsed_time(t1) / 20 * 1e3:.1f} us per launch by events (eager), {(x.numel() + y.numel() + bank.numel()) * 4 / 1e6:.1f} MB')
fn = hip.lib.hs_debug_read_stamps
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
n = 8192 * 32
buf = np.zeros(n, dtype=np.int64)
assert fn(buf.ctypes.data, n) == 0
st = buf.reshape(8192, 32)
st = st[st[:, 24] > 0]
print(f'{len(st)} workgroups stamped')
labels = {0: 'A fragments requested', 1: 'two tiles requested', 2: 'A masked (A landed)', 20: 'steady loop done', 24: 'last tiles + stores issued'}
for k in range(3, 14):
    labels[k] = f'tile {4 * (k - 3)} done + refill issued'
prev = None
for k in sorted(labels):
    col = st[:, k]
    if (col == 0).all():
        continue
    rel = col - st[:, 0]
    d = (col - st[:, prev]) if prev is not None else rel
    print(f'  stamp {k:2d} {labels[k]:34s} since start: mean {rel.mean():8.0f} | phase: mean {d.mean():7.0f} min {d.min():7.0f} max {d.max():7.0f}')
    prev = k
life = st[:, 24] - st[:, 0]
span = st[:, 29] - st[:, 30].min()                      # s_memrealtime (100 MHz) at the end, relative to the first workgroup's start
print(f'workgroup life: mean {life.mean():.0f} cycles; launch span by s_memrealtime: {span.max() / 100.0:.1f} us; starts spread over {(st[:, 30] - st[:, 30].min()).max() / 100.0:.1f} us')
hw = st[:, 31]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
cu_key = (xcc << 16) | (((hwid >> 13) & 0x7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)
per_cu = collections.Counter(cu_key.tolist())
print(f'{len(per_cu)} CUs used, workgroups per CU: min {min(per_cu.values())} max {max(per_cu.values())}')
