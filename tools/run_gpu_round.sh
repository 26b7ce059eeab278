set -o pipefail
for cfg in "0 0" "4 128" "4 192" "4 384" "3 128" "0 0"; do
  set -- $cfg
  SHM_WGRAD_BF16_WIDE=$1 SHM_WGRAD_BLOCKS=$2 timeout -k 10 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/r4_ab.json 2>gpurun_out/r4_ab.err || { tail -3 gpurun_out/r4_ab.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ab.json") if l.startswith("{")][0])
print("bf16 wide=$1 blocks=$2", j["ms_per_step"], j["value"])
PY
done
