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


def _demangle(name):
    """rocprofv3 leaves symbols with __bf16 template arguments mangled (its demangler does not know DF16b).
    Enough of the Itanium grammar for this library's kernels: _Z<len><name>I<template args>E<function args>."""
    m = re.match(r"_Z(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return base
    rest, args = rest[1:], []
    while rest and not rest.startswith("E"):
        if rest.startswith("DF16b"):
            args.append("__bf16")
            rest = rest[5:]
        elif rest[0] == "f":
            args.append("float")
            rest = rest[1:]
        elif rest.startswith("Lb"):
            args.append("true" if rest[2] == "1" else "false")
            rest = rest[4:]
        elif rest.startswith("Li"):
            j = rest.index("E")
            args.append(rest[2:j])
            rest = rest[j + 1:]
        else:
            return name
    return f"{base}<{', '.join(args)}>"


def norm(name):
    if name.startswith("_Z"):
        name = _demangle(name)
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


def main():
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


if __name__ == "__main__":
    main()
