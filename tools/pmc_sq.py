"""Per-kernel SQ counters from one rocprofv3 --pmc pass (CSV output):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \\
              --output-format csv -d D -- python3 bench.py --serialize ...
    python tools/pmc_sq.py D > profiles/<round>_sq_pmc.json

mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): share of the busy-CU time in which the MFMA
pipes (4 per CU) were busy; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE."""
import csv
import glob
import json
import sys
from collections import defaultdict

from pmc_traffic import norm

acc = defaultdict(lambda: defaultdict(float))
disp = defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = norm(row["Kernel_Name"])
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        disp[k].add(row["Dispatch_Id"])
out = {}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CU_CYCLES", 0)):
    if v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
        continue
    out[k] = {"launches": len(disp[k]),
              "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * v["SQ_BUSY_CU_CYCLES"]),
              "lds_bank_conflict_frac": v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0), 1.0)}
json.dump(out, sys.stdout, indent=1)
