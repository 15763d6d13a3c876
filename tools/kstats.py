"""Prints the hs:: kernels of a rocprofv3 kernel_stats.csv:  name  calls  avg_us  (sorted by total time)."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 18]:
    n = r['Name']
    if flt and flt not in n:
        continue
    print(f"{float(r['AverageNs']) / 1e3:9.2f} us x {int(r['Calls']):5d}  {float(r['Percentage']):5.1f}%  {n[:110]}")
