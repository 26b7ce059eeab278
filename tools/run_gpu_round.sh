set -o pipefail
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_dist_gpu.py tests/test_variants_gpu.py -m gpu -q -s -k "config3 or config4 or ranks or nan" > gpurun_out/r4_t5.log 2>&1
echo "pytest rc=$?"; grep -v "amdgpu.ids" gpurun_out/r4_t5.log | tail -40
