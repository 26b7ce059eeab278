set -o pipefail
timeout -k 10 1100 python -m pytest tests -m gpu -q -x -k "wgrad or fold or reproducible or parity or config or ranks or golden" > gpurun_out/r4_t8.log 2>&1
rc=$?
echo "pytest rc=$rc"; grep -v amdgpu.ids gpurun_out/r4_t8.log | tail -12
if [ $rc -eq 0 ]; then
timeout -k 10 300 python tools/bench_wgrad_bf16.py 40,256,64,64,1 40,128,128,128,1 40,64,256,256,1 40,128,128,256,2 > gpurun_out/r4_wg2.txt 2>&1; cat gpurun_out/r4_wg2.txt | grep -v amdgpu
fi
