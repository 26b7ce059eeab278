"""Distil a rocprofv3 kernel_stats csv: per-kernel calls, ms per step, average us (steps = 3 in the profile scripts)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print(f"total {tot:.3f} ms/step")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"{re.sub(r'[(].*[)]$', '', r['Name'])[:100]:100s} {int(r['Calls']):5d} {float(r['TotalDurationNs']) / steps / 1e6:8.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us")
