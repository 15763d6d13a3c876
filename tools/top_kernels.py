import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per frame", tot / n / 1e6, "kernels per frame", sum(int(r["Calls"]) for r in rows) / n)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%9.1f us/frame %6.1f calls avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / n / 1e3, int(r["Calls"]) / n,
                                                        float(r["AverageNs"]) / 1e3, r["Name"][:120]))
