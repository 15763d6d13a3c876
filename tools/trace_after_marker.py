"""Per-kernel stats of the dispatches AFTER the last marker kernel in a rocprofv3 kernel_trace.csv."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = max(i for i, r in enumerate(rows) if 'scan' in r['Kernel_Name'].lower() or 'cumsum' in r['Kernel_Name'].lower())
rows = rows[last + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg[r['Kernel_Name']]
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
span = int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])
print('dispatches/frame %.1f   kernel time/frame %.1f us   wall span/frame %.1f us' % (len(rows) / n, tot / n / 1e3, span / n / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%9.1f us/frame %6.1f calls avg %7.1f us  %s' % (v[1] / n / 1e3, v[0] / n, v[1] / v[0] / 1e3, k[:110]))
if len(sys.argv) > 4:
    sub = sys.argv[4]
    ds = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if sub in r['Kernel_Name']]
    per = int(len(ds) / n)
    print(sub, 'per-call us (first frame):', [round(d, 1) for d in ds[:per]])
