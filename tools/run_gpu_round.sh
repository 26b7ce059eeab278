set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=5 --deselect tests/test_bf16_gpu.py --deselect tests/test_attention_gpu.py --deselect tests/test_bench_gpu.py --deselect tests/test_dist_gpu.py > gpurun_out/r4_t7.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_t7.log | tail -30
