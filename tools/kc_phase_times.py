"""Per-phase cycle counts of hs_k1_chain_fwd (dev build 'stamps_kc' of tools/build_variants.py):
    HS_K1_CHAIN=1 HS_HIP_LIB=hyperseg_amd/lib/libhyperseg_hip_stamps_kc.so python tools/kc_phase_times.py [M|Sc]"""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _workload import decoder_workload
import hyperseg_amd._hip as hip

name = sys.argv[1] if len(sys.argv) > 1 else 'M'
dec, pyr, head = decoder_workload(name)
dec.chain_k1 = True
for _ in range(4):
    dec(pyr, head)
torch.cuda.synchronize()
fn = hip.lib.hs_debug_read_stamps
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
n = 8192 * 32
buf = np.zeros(n, dtype=np.int64)
assert fn(buf.ctypes.data, n) == 0
st = buf.reshape(8192, 32)
st = st[st[:, 24] > 0]
labels = {0: 'start', 1: 'all DMA issued (groups 0 + 1)', 2: 'group 0 landed (vmcnt(10))', 3: 'barrier, generation read', 4: 'level 0 products + barrier + publish',
          5: 'barrier (group 1 landed)', 6: 'gather 0 -> 1 (neighbours\' level 0)', 7: 'stage 1 built (2 barriers)', 8: 'level 1 products + publish',
          9: 'gather 1 -> 2', 10: 'stage 2 built (2 barriers)', 11: 'level 2 products + publish / stores', 12: 'gather 2 -> 3 (6 x 6 window)',
          13: 'halo tile built (2 barriers)', 14: 'pw1 + bn1 + relu6 -> h1', 15: 'depthwise + bn2 + relu6 -> h2', 24: 'pw3 + bn3 + stores (end)'}
print(f'{name}: {len(st)} workgroups stamped; shader clock cycles (wave 0 of every workgroup)')
prev = None
for k in sorted(labels):
    col = st[:, k]
    if (col == 0).all():
        continue
    rel = col - st[:, 0]
    d = (col - st[:, prev]) if prev is not None else rel
    print(f'  stamp {k:2d} {labels[k]:44s} since start: mean {rel.mean():8.0f} max {rel.max():8.0f} | phase: mean {d.mean():7.0f} min {d.min():7.0f} max {d.max():7.0f}')
    prev = k
t_first, t_last = st[:, 30].min(), st[:, 29].max()
print(f'  first start -> last end: {(t_last - t_first) / 100.0:.2f} us (100 MHz realtime counter); start skew {(st[:, 30].max() - t_first) / 100.0:.2f} us')
