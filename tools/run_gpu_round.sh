set -o pipefail
timeout -k 10 600 python -m pytest tests/test_rgb_gpu.py -m gpu -q -x > gpurun_out/r4_rgb.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_rgb.log | tail -25
