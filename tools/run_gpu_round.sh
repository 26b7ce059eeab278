set -o pipefail
run() { tag=$1; shift; timeout -k 10 400 python bench.py "$@" > gpurun_out/end4_$tag.json 2> gpurun_out/end4_$tag.err || { echo FAIL $tag; tail -3 gpurun_out/end4_$tag.err; exit 1; }
python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/end4_$tag.json") if l.startswith("{")][0])
r=j.get("roofline",{})
print("$tag", j["ms_per_step"], j["value"], j["unit"], "roofline", r.get("kernel"), r.get("frac"), "conv", r.get("whole_step_conv_tflops"), "cpu", (j.get("cpu_baseline") or {}).get("value"), "north*", (j.get("north_star_block") or {}).get("us"))
PY
}
run f32 --steps 10 --warmup 3
run bf16 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline
run s512_b4_bf16 --dtype bf16 --image-size 512 --batch 4 --steps 8 --warmup 2 --no-cpu-baseline
run b32_bf16 --dtype bf16 --batch 32 --steps 8 --warmup 2 --no-cpu-baseline
run s512_b4_f32 --image-size 512 --batch 4 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer
run b32_f32 --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timer
