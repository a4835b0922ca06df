"""Print the top rows of a rocprofv3 kernel_stats.csv with min / max: python tools/show_stats.py FILE [N] [substring ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
keys = sys.argv[3:]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel ms {tot / 1e6:.1f}")
for r in (rows if keys else rows[:n]):
    if keys and not any(k in r["Name"] for k in keys):
        continue
    print(f"{r['Name'][:78]:78s} {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:7.1f} min {float(r['MinNs']) / 1e3:7.1f} max {float(r['MaxNs']) / 1e3:7.1f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
