set -o pipefail
timeout -k 10 900 python -m pytest tests/test_step_gpu.py tests/test_train_loop_gpu.py tests/test_bf16_gpu.py tests/test_dist_gpu.py tests/test_norm_fold_gpu.py -m gpu -q -x > gpurun_out/r4_rs.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_rs.log | tail -4
[ $rc -eq 0 ] || exit 1
for dt in bf16 f32; do
for v in 0 1 0 1; do
  SHM_WGRAD_REDUCE_STREAM=$v timeout -k 10 300 python bench.py --dtype $dt --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/r4_ab_$v.json 2>gpurun_out/r4_ab_$v.err || exit 1
  python - <<PY
import json
j=json.loads([l for l in open("gpurun_out/r4_ab_$v.json") if l.startswith("{")][0])
print("$dt reduce-stream=$v", j["ms_per_step"], j["value"])
PY
done; done
