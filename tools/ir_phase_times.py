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
# round 6: the launch on the chip-wide 100 MHz clock (s_memrealtime is the same counter on every XCD): when workgroups start and end
rs, re = st[:, 30].astype(np.float64), st[:, 29].astype(np.float64)
r0 = rs.min()
q = lambda a, p: np.percentile(a, p)     # noqa: E731
print(f'  real time (us since the first workgroup started): starts p0 {0.0:.2f} p10 {q(rs - r0, 10) / 100:.2f} p50 {q(rs - r0, 50) / 100:.2f} '
      f'p90 {q(rs - r0, 90) / 100:.2f} p100 {(rs - r0).max() / 100:.2f} | ends p0 {(re - r0).min() / 100:.2f} p10 {q(re - r0, 10) / 100:.2f} '
      f'p50 {q(re - r0, 50) / 100:.2f} p90 {q(re - r0, 90) / 100:.2f} p100 {(re - r0).max() / 100:.2f}')
first_per_cu = sorted(min(a for a, _ in iv) for iv in per_cu.values())
order = np.argsort(rs)
print(f'  the 256th workgroup (of {len(rs)}) started {(np.sort(rs)[min(255, len(rs) - 1)] - r0) / 100:.2f} us after the first; the last {(rs.max() - r0) / 100:.2f} us')
# round 6: who is slow?  lifetimes by the order in which a CU's workgroups started, and the phase sums of the two classes
rank = np.zeros(len(st), dtype=np.int64)
for k in per_cu:
    ids = [i for i in range(len(st)) if int(cu_key[i]) == k]
    for r, i in enumerate(sorted(ids, key=lambda i: st[i, 0])):
        rank[i] = r
life = (st[:, 24] - st[:, 0]).astype(np.float64)
for r in sorted(set(rank.tolist()))[:4]:
    m = rank == r
    print(f'  workgroups that started {r + 1}. on their CU: n {int(m.sum())}, lifetime mean {life[m].mean():.0f} p10 {np.percentile(life[m], 10):.0f} p90 {np.percentile(life[m], 90):.0f}; '
          f'start offset to the CU\'s first: mean {np.mean([st[i, 0] - min(a for a, _ in per_cu[int(cu_key[i])]) for i in np.nonzero(m)[0]]):.0f} cycles; '
          f'prologue (-> stamp 7) {np.mean(st[m, 7] - st[m, 0]):.0f}, chunk loop + epilogue {np.mean(st[m, 24] - st[m, 7]):.0f}')
print('  lifetime by XCD: ' + ', '.join(f'{int(x)}: {life[xcc == x].mean():.0f}' for x in sorted(set(xcc.tolist()))))
slow = life >= np.percentile(life, 90)
print(f'  slowest 10 %: started {np.mean(rank[slow] == 0) * 100:.0f} % first on their CU; prologue {np.mean(st[slow, 7] - st[slow, 0]):.0f} vs all {np.mean(st[:, 7] - st[:, 0]):.0f}; '
      f'chunks {np.mean(st[slow, 24] - st[slow, 7]):.0f} vs all {np.mean(st[:, 24] - st[:, 7]):.0f}')
