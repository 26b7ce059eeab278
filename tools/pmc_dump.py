"""Per-kernel means of every counter in a rocprofv3 --pmc output directory (CSV), kernels matching a substring:
    python tools/pmc_dump.py <dir> [substring]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if pat not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
for k, v in acc.items():
    n = len(disp[k])
    print(f"{k}  launches {n}")
    for c, val in sorted(v.items()):
        print(f"    {c:32s} {val / n:16.1f}")
