set -o pipefail
timeout -k 10 900 python -m pytest tests/test_dist_gpu.py tests/test_bench_gpu.py -m gpu -q -x > gpurun_out/r4_t11.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_t11.log | tail -25
