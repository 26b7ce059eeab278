set -o pipefail
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_step_gpu.py -m gpu -q -x -k "first_layer or golden or parity" > gpurun_out/r4_t10.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_t10.log | tail -6
echo "--- new"; timeout -k 10 120 python tools/probes/bench_sum1.py 2>&1 | grep -v amdgpu
echo "--- old"; SHM_LIB_PATH=build_ab/lib_old.so timeout -k 10 120 python tools/probes/bench_sum1.py 2>&1 | grep -v amdgpu
