set -o pipefail
for cfg in "0 0" "1 128" "1 256" "1 512" "0 0" "1 128" "1 256"; do
  set -- $cfg
  SHM_GSUM=$1 SHM_GSUM_MINC=$2 timeout -k 10 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/r4_ab.json 2>gpurun_out/r4_ab.err || { tail -5 gpurun_out/r4_ab.err; exit 1; }
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ab.json") if l.startswith("{")][0])
print("bf16 gsum=$1 minc=$2", j["ms_per_step"], j["value"])
PY
done
