"""Per-phase cycle counts of the fused inverted-residual kernels (dev builds of tools/build_variants.py):
    HS_IR_MATH=auto HS_HIP_LIB=hyperseg_amd/lib/libhyperseg_hip_stamps_irc.so python tools/ir_phase_times.py [M|S|Sc]     (hs_patch_irc.hip)
    HS_STAMP_LABELS=fused HS_HIP_LIB=hyperseg_amd/lib/libhyperseg_hip_stamps.so python tools/ir_phase_times.py [M|S|Sc|L]  (hs_patch_ir_fused.hip)"""
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
for _ in range(3):
    dec(pyr, head)
torch.cuda.synchronize()
fn = hip.lib.hs_debug_read_stamps
fn.argtypes, fn.restype = [C.c_void_p, C.c_int], C.c_int
n = 8192 * 32
buf = np.zeros(n, dtype=np.int64)
assert fn(buf.ctypes.data, n) == 0
st = buf.reshape(8192, 32)
live = st[:, 24] > 0
st = st[live]
print(f'{name}: {live.sum()} workgroups stamped (the LAST inverted-residual launch of the decoder)')
if os.environ.get('HS_STAMP_LABELS', 'irc') == 'irc':          # hs_patch_irc.hip ('stamps_irc' build)
    labels = {0: 'start', 1: 'tile loads issued, BN rows stored', 2: 'tiles + tap tables -> scratch, bank DMA issued', 3: 'barrier',
              4: 'B fragments built', 5: 'barrier (DMA landed)', 6: 'bank split LDS -> LDS', 7: 'barrier', 9: 'pw1(0) + barrier',
              24: 'epilogue stores issued'}
    for c in range(3):
        labels[10 + 4 * c] = f'dw[{c}{"+" if c == 2 else ""}]'
        labels[11 + 4 * c] = 'barrier'
        labels[12 + 4 * c] = 'pw3'
        labels[13 + 4 * c] = 'pw1(next) + barrier'
else:                                                          # hs_patch_ir_fused.hip ('stamps' build)
    labels = {0: 'start', 1: 'loads issued + LDS stores', 2: 'barrier (window in LDS)', 3: 'B fragments done', 4: 'pw1(0) + barrier',
              24: 'epilogue stores issued'}
    for c in range(4):
        labels[5 + 4 * c] = f'dw[{c}{"+" if c == 3 else ""}]'
        labels[6 + 4 * c] = 'barrier'
        labels[7 + 4 * c] = 'pw3'
        labels[8 + 4 * c] = 'pw1(next) + barrier'
t0 = st[:, 0].min()
prev = None
for k in sorted(labels):
    col = st[:, k]
    if (col == 0).all():
        continue
    rel = col - st[:, 0]
    d = (col - st[:, prev]) if prev is not None else rel
    print(f'  stamp {k:2d} {labels[k]:46s} since start: mean {rel.mean():9.0f}  | phase: mean {d.mean():8.0f} min {d.min():8.0f} max {d.max():8.0f}')
    prev = k
# residency: workgroups per CU over time, from (XCC id, HW_ID) of wave 0 of every workgroup
hw = st[:, 31]
xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
cu_key = (xcc << 16) | (((hwid >> 13) & 0x7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 0xf)      # xcc, se, sh, cu
import collections
per_cu = collections.defaultdict(list)
for k, a, b in zip(cu_key, st[:, 0], st[:, 24]):
    per_cu[int(k)].append((int(a), int(b)))
busy = []
for k, iv in per_cu.items():
    ev = sorted([(a, 1) for a, _ in iv] + [(b, -1) for _, b in iv])
    cur, last, area, peak = 0, ev[0][0], 0, 0
    for t, d in ev:
        area += cur * (t - last); last = t; cur += d; peak = max(peak, cur)
    span = ev[-1][0] - ev[0][0]                     # per CU: s_memtime is not synchronised across XCDs
    gaps = sorted(a for a, _ in iv)
    busy.append((area / span, peak, len(iv), span))
print(f'  CUs seen: {len(per_cu)}; workgroups per CU: mean {np.mean([b[2] for b in busy]):.1f}; resident workgroups per CU over the launch: '
      f'mean {np.mean([b[0] for b in busy]):.2f}, peak {max(b[1] for b in busy)}; per-CU span mean {np.mean([b[3] for b in busy]):.0f} cycles')
real = (st[:, 29] - st[:, 30]).astype(np.float64)          # s_memrealtime: constant 100 MHz
cyc = (st[:, 24] - st[:, 0]).astype(np.float64)
print(f'  shader clock while these workgroups ran (cycle counter / 100 MHz real-time counter): {np.median(cyc / real) * 0.1:.2f} GHz; '
      f'workgroup lifetime {np.median(real) / 100:.1f} us')
print(f'  workgroup lifetime: mean {(st[:, 24] - st[:, 0]).mean():.0f}, p10 {np.percentile(st[:, 24] - st[:, 0], 10):.0f}, p90 {np.percentile(st[:, 24] - st[:, 0], 90):.0f}')
print(f'  workgroup start spread: {(st[:, 0] - t0).max()} cycles; total mean {(st[:, 24] - st[:, 0]).mean():.0f}, '
      f'first start -> last end {st[:, 24].max() - t0}')
