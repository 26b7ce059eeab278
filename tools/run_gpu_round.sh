set -o pipefail
for dt in f32 bf16; do
for v in base pipe base pipe; do
  L=""; [ $v != base ] && L=$PWD/build_ab/lib_$v.so
  SHM_LIB_PATH=$L timeout -k 10 300 python bench.py --dtype $dt --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r4_ab.json 2>gpurun_out/r4_ab.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ab.json") if l.startswith("{")][0])
h=j.get("roofline_hbm",{}).get("passes",{})
print("$dt $v", j["ms_per_step"], j["value"], {k:(v["ms_per_step"], v["GBps"]) for k,v in h.items() if "bwd" in k})
PY
done; done
