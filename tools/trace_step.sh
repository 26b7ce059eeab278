#!/bin/bash
# rocprofv3 kernel trace (timestamps) of the PRODUCTION step (two streams), for tools/timeline.py
#   SHM_TREE_SHA=<git sha> tools/trace_step.sh <tag> [bench.py arguments ...]   -> gpurun_out/trace_<tag>/distilled/<tag>_timeline.json
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/trace_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$out/t" -- python3 "$root/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-extra-configs $* > "$out/log.txt" 2>&1
cd "$root"
f=$(find "$out/t" -name '*kernel_trace.csv' | head -1)
export SHM_CSRC_SHA=$(cat shmgan_amd/csrc/*.hip shmgan_amd/csrc/*.h | sha256sum | cut -c1-16)
mkdir -p "$out/distilled"
python3 tools/timeline.py "$f" "$out/distilled/${tag}_timeline.json" > "$out/timeline.txt"
cp "$out/distilled/${tag}_timeline.json" profiles/ 2>/dev/null || true
# keep only the distilled timeline (the raw trace is tens of MB)
python3 - "$f" "$out/trace_small.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
with open(sys.argv[2], "w") as f:
    for r in rows:
        f.write(f'{int(r["Start_Timestamp"]) - t0},{int(r["End_Timestamp"]) - t0},{r["Queue_Id"]},{r["Kernel_Name"][:60]}\n')
PY
rm -rf "$out/t"
echo "[$tag] trace done"
