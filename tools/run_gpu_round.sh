set -o pipefail
for g in 0 1 0 1; do
  SHM_GSUM=$g timeout -k 10 200 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/r4_gsum_$g.json 2>gpurun_out/r4_gsum_$g.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_gsum_$g.json") if l.startswith("{")][0])
print("bf16 gsum=$g", j["ms_per_step"], j["value"])
PY
done
