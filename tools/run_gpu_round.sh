set -o pipefail
export SHM_TREE_SHA=9a3106c
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/end4_default.json 2> gpurun_out/end4_default.err; echo "default rc=$?"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --dtype bf16 --no-cpu-baseline > gpurun_out/end4_bf16.json 2> gpurun_out/end4_bf16.err; echo "bf16 rc=$?"
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --dtype bf16 --image-size 512 --batch 4 --no-cpu-baseline > gpurun_out/end4_s512.json 2> gpurun_out/end4_s512.err; echo "s512 rc=$?"
timeout -k 10 200 python bench.py --steps 8 --warmup 3 --dtype bf16 --batch 32 --no-cpu-baseline > gpurun_out/end4_b32.json 2> gpurun_out/end4_b32.err; echo "b32 rc=$?"
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --image-size 512 --batch 4 --no-cpu-baseline --no-kernel-timer > gpurun_out/end4_s512_f32.json 2> /dev/null; echo "s512 f32 rc=$?"
timeout -k 10 200 python bench.py --steps 8 --warmup 3 --batch 32 --no-cpu-baseline --no-kernel-timer > gpurun_out/end4_b32_f32.json 2> /dev/null; echo "b32 f32 rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --same-device --backend gloo --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > gpurun_out/end4_gloo2.json 2> gpurun_out/end4_gloo2.err; echo "gloo2 rc=$?"
python - <<'PY'
import json
for f in ["default","bf16","s512","b32","s512_f32","b32_f32","gloo2"]:
    try:
        j=json.loads(open(f"gpurun_out/end4_{f}.json").read().strip().splitlines()[-1])
        r=j.get("roofline",{})
        print(f, j["ms_per_step"], j["value"], r.get("kernel"), r.get("frac"), r.get("whole_step_conv_tflops"), j.get("north_star_block",{}).get("us"), j.get("comm"), j.get("cpu_baseline",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
