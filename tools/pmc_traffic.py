"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (CSV output), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950:

    bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024       (FETCH_SIZE counts half of 16-byte/lane streaming reads)

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -- python3 bench.py --serialize ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d B -- python3 bench.py --serialize ...
    python tools/pmc_traffic.py A B > profiles/<round>_traffic_pmc.json
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def norm(name):
    return re.sub(r"\(.*\)$", "", name.replace("void ", "")).strip()


def collect(d):
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = norm(row["Kernel_Name"])
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in disp.items()}


a, na = collect(sys.argv[1])
b, nb = collect(sys.argv[2])
out = {}
for k in sorted(a, key=lambda k: -a[k].get("FETCH_SIZE", 0)):
    if k not in b or a[k].get("FETCH_SIZE", 0) < 1e5:
        continue
    n = na[k]
    fetch, write = a[k]["FETCH_SIZE"], b[k].get("WRITE_SIZE", 0.0)
    hit, miss = b[k].get("TCC_HIT_sum", 0.0), b[k].get("TCC_MISS_sum", 0.0)
    out[k] = {"launches": n, "fetch_kb_raw": fetch, "write_kb": write,
              "hbm_bytes_per_launch": (2 * fetch + write) * 1024 / n,
              "l2_hit": hit / max(hit + miss, 1.0)}
json.dump(out, sys.stdout, indent=1)
