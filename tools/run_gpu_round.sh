set -o pipefail
timeout -k 10 900 python -m pytest tests/test_rgb_gpu.py tests/test_step_gpu.py tests/test_bf16_gpu.py tests/test_train_loop_gpu.py tests/test_attention_gpu.py -m gpu -q -x > gpurun_out/r4_rgb.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_rgb.log | tail -4
[ $rc -eq 0 ] || exit 1
for dt in f32 bf16; do
  for lay in staging compact staging compact; do
    SHM_D_INPUT=$lay timeout -k 10 200 python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r4_rgb_${dt}_${lay}.json 2>gpurun_out/r4_rgb_${dt}_${lay}.err || exit 1
    python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_rgb_${dt}_${lay}.json") if l.startswith("{")][0])
print("$dt $lay", j["ms_per_step"], j["value"])
PY
  done
done
