#!/bin/bash
# rocprofv3 kernel-trace stats of one bench configuration (serialized step), distilled to profiles/<tag>_kernel_stats_serialize.csv
#   tools/stats_only.sh <tag> [bench.py arguments ...]
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
args="--serialize --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" $args > "$out/stats.log" 2>&1
cd "$root"
mkdir -p "$out/distilled"
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "$out/distilled/${tag}_kernel_stats_serialize.csv"
find "$out" -name '*.csv' -size +2M -delete
echo "[$tag] stats pass done"
