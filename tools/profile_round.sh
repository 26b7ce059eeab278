#!/bin/bash
# rocprofv3 evidence for one bench configuration (run ON the GPU box, from the repo root):
#   SHM_TREE_SHA=<git sha> tools/profile_round.sh <tag> [bench.py arguments ...]
# four passes of the same command (kernel-trace stats; PMC FETCH_SIZE; PMC WRITE_SIZE + L2 hit/miss; SQ counters), each in its
# own run as the MI355X guide prescribes (no --pmc together with --stats), distilled into profiles/<tag>_*.{csv,json}.
set -e -o pipefail
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
args="--serialize --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-configs $*"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/bench.py" $args > "$out/stats.log" 2>&1
echo "[$tag] stats pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $args > "$out/fetch.log" 2>&1
echo "[$tag] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$out/write" -- python3 "$root/bench.py" $args > "$out/write.log" 2>&1
echo "[$tag] WRITE_SIZE pass done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/sq" -- python3 "$root/bench.py" $args > "$out/sq.log" 2>&1
echo "[$tag] SQ pass done"
cd "$root"
# what tree the numbers belong to: the git commit (handed in by the caller: the GPU box has no .git) and a hash of the kernel sources as
# they are ON the box -- anyone can recompute the latter:  cat shmgan_amd/csrc/*.hip shmgan_amd/csrc/*.h | sha256sum | cut -c1-16
export SHM_CSRC_SHA=$(cat shmgan_amd/csrc/*.hip shmgan_amd/csrc/*.h | sha256sum | cut -c1-16)
cp "$(find "$out/stats" -name '*kernel_stats.csv' | head -1)" "profiles/${tag}_kernel_stats_serialize.csv"
python3 tools/pmc_traffic.py "$out/fetch" "$out/write" > "profiles/${tag}_traffic_pmc.json"
(cd tools && python3 pmc_sq.py "$out/sq") > "profiles/${tag}_sq_pmc.json"
python3 - "$tag" "$*" <<'PY'
import json, os, sys
tag, args = sys.argv[1], sys.argv[2]
meta = {"tree": os.environ.get("SHM_TREE_SHA", "unknown"), "csrc_sha16": os.environ["SHM_CSRC_SHA"],
        "command": "bench.py --serialize --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-configs " + args,
        "passes": ["--kernel-trace --stats", "--pmc FETCH_SIZE", "--pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
                   "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"]}
for kind in ("traffic_pmc", "sq_pmc"):
    p = f"profiles/{tag}_{kind}.json"
    j = json.load(open(p))
    j = {"_meta": meta, **j}
    json.dump(j, open(p, "w"), indent=1)
json.dump(meta, open(f"profiles/{tag}_meta.json", "w"), indent=1)
PY
# gpurun merges only gpurun_out/ back: carry the distilled files there too (copy them into profiles/ afterwards)
mkdir -p "$out/distilled" && cp profiles/${tag}_kernel_stats_serialize.csv profiles/${tag}_traffic_pmc.json profiles/${tag}_sq_pmc.json profiles/${tag}_meta.json "$out/distilled/"
# keep the merged-back scratch small: the raw counter CSVs are tens of MB
find "$out" -name '*.csv' -size +2M -delete
echo "[$tag] distilled into profiles/${tag}_*"
