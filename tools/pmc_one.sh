#!/bin/bash
# HBM-side traffic of the fp32 weight-gradient kernels only (two PMC passes of a short serialized bench): prints bytes per launch
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/pmc_one; rm -rf "$out"; mkdir -p "$out"
args="--serialize --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timer $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $args > "$out/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/write" -- python3 "$root/bench.py" $args > "$out/write.log" 2>&1
cd "$root"
python3 tools/pmc_traffic.py "$out/fetch" "$out/write" > "$out/traffic.json"
python3 - "$out/traffic.json" <<'PY'
import json, sys
t = json.load(open(sys.argv[1]))
for k, v in t.items():
    if "wgrad" in k:
        print(k[:70], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
find "$out" -name '*.csv' -size +1M -delete
