"""In-order list of the dispatches of ONE graph-replayed frame from a rocprofv3 kernel_trace.csv (after the marker scan
kernel of tools/prof_graph.py), averaged by position over the n replays: duration, idle gap before it, grid, name.
    python tools/frame_sequence.py <kernel_trace.csv> <n_replays> > profiles/<name>.txt"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'scan' in r['Kernel_Name'].lower() or 'cumsum' in r['Kernel_Name'].lower())
rows = rows[last + 1:]
per = len(rows) // n
assert per * n == len(rows), (len(rows), n)
frames = [rows[i * per:(i + 1) * per] for i in range(n)]
for f in frames:
    assert [r['Kernel_Name'] for r in f] == [r['Kernel_Name'] for r in frames[0]]
tot_d = tot_g = 0.0
print(f'{per} dispatches per frame, averaged over {n} replays')
print(' idx   dur_us  gap_us  grid(x,y,z)/wg            kernel')
for i in range(per):
    d = sum(int(f[i]['End_Timestamp']) - int(f[i]['Start_Timestamp']) for f in frames) / n / 1e3
    g = sum(int(f[i]['Start_Timestamp']) - int(f[i - 1]['End_Timestamp']) for f in frames) / n / 1e3 if i else 0.0
    r = frames[0][i]
    wg = int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
    grid = '%d,%d,%d' % tuple(int(r[f'Grid_Size_{a}']) // max(int(r[f'Workgroup_Size_{a}']), 1) for a in 'XYZ')
    tot_d += d
    tot_g += g
    print(f'{i:4d} {d:8.2f} {g:7.2f}  {grid + "/" + str(wg):24s}  {r["Kernel_Name"][:90]}')
span = sum(int(f[-1]['End_Timestamp']) - int(f[0]['Start_Timestamp']) for f in frames) / n / 1e3
print(f'kernel time {tot_d:.1f} us + gaps {tot_g:.1f} us = span {span:.1f} us per frame')
