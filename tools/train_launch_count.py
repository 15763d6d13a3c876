"""Launches and kernel time of ONE config-5 training step from two rocprofv3 --kernel-trace --stats runs of tools/train_step_time.py
with different step counts: (calls_b - calls_a) / (steps_b - steps_a) per kernel -- model set-up (weights, the encoder pass that
produces the features) cancels out.
    python tools/train_launch_count.py <stats_a.csv> <steps_a> <stats_b.csv> <steps_b> [rows]"""
import csv
import sys

fa, na, fb, nb = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
top = int(sys.argv[5]) if len(sys.argv) > 5 else 0


def load(path):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(path))}


a, b = load(fa), load(fb)
rows = []
for name, (cb, tb) in b.items():
    ca, ta = a.get(name, (0, 0.0))
    per, us = (cb - ca) / (nb - na), (tb - ta) / (nb - na) / 1e3
    if abs(per) > 1e-9:
        rows.append((per, us, name))
own = [r for r in rows if 'hs::' in r[2][:12]]
stock = [r for r in rows if 'hs::' not in r[2][:12]]
print(f'per step: {sum(r[0] for r in rows):.1f} launches, {sum(r[1] for r in rows):.1f} us of kernel time '
      f'| own {sum(r[0] for r in own):.1f} launches {sum(r[1] for r in own):.1f} us | stock {sum(r[0] for r in stock):.1f} launches {sum(r[1] for r in stock):.1f} us')
for per, us, name in sorted(rows, key=lambda r: -r[1])[:top]:
    print(f'{per:6.1f} x {us / per:7.2f} us = {us:7.1f}  {name[:110]}')
